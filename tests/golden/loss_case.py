"""Seeded inputs of the loss-path golden case (shared by make_golden.gen_losses and tests/test_losses.py):
head outputs of 6 decoder layers for bs = 2 and ragged ground truth in the reference's data format
(projects/configs/hipad_b2d_stage2.py:516-523 keys).  Sample 0 has 17 boxes / 7 map lines, sample 1 has 5 boxes
and NO map line; one box is a traffic cone (class-wise regression weights), one has a NaN velocity (NaN-aware
weights), half of the class logits sit under cls_threshold_to_reg."""
import math

import torch

from seeded import seeded

BS, LAYERS = 2, 6
N_DET, N_MAP, N_PLAN = 900, 100, 480
DET_CLS, MAP_CLS, MODES, TS = 9, 4, 6, 6
NUM_PERM = 38


def head_outputs(requires_grad=False):
    def t(shape, seed, scale=1.0, shift=0.0):
        x = seeded((BS,) + shape, seed, scale) + shift
        return x.requires_grad_(True) if requires_grad else x
    det = dict(classification=[t((N_DET, DET_CLS), 100 + i, 1.0, -4.5) for i in range(LAYERS)],
               prediction=[t((N_DET, 11), 110 + i, 2.0) for i in range(LAYERS)],
               quality=[t((N_DET, 2), 120 + i) for i in range(LAYERS)])
    mp = dict(classification=[t((N_MAP, MAP_CLS), 130 + i, 1.0, -3.5) for i in range(LAYERS)],
              prediction=[t((N_MAP, 40), 140 + i, 8.0) for i in range(LAYERS)],
              quality=[None] * LAYERS)
    ego = dict(classification=[None] * LAYERS, prediction=[None] * LAYERS,
               status=[t((1, 6), 150 + i) for i in range(LAYERS)])
    plan = dict(classification=[t((1, N_PLAN), 160 + i) for i in range(LAYERS)],
                prediction=[t((1, N_PLAN, TS, 2), 170 + i) for i in range(LAYERS)], status=[None] * LAYERS)
    motion = dict(classification=[t((N_DET, MODES), 180 + i) for i in range(LAYERS)],
                  prediction=[t((N_DET, MODES, TS, 2), 190 + i) for i in range(LAYERS)])
    scenes = dict(scenes_latent_tokens=[], scenes_latent_embeds=[], scenes_future_tokens=[], scenes_future_embeds=[])
    return det, mp, ego, plan, motion, scenes


def ground_truth():
    g = torch.Generator().manual_seed(777)
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    u = lambda *s: torch.rand(*s, generator=g)  # noqa: E731
    data = {}
    boxes, labels, trajs, masks = [], [], [], []
    for n in (17, 5):
        b = torch.cat([r(n, 3) * torch.tensor([10.0, 20.0, 1.0]), u(n, 3) * 3 + 0.5, (u(n, 1) * 2 - 1) * math.pi, r(n, 2) * 2], dim=1)
        lab = torch.randint(0, DET_CLS, (n,), generator=g)
        boxes.append(b); labels.append(lab)
        trajs.append(r(n, TS, 2)); masks.append((u(n, TS) > 0.3).float())
    labels[0][3] = 5             # traffic_cone: class-wise regression weights
    boxes[0][4, 7] = float("nan")  # unknown velocity: weight 0 and target 0 for that component
    masks[0][2] = 0.0            # an agent without any valid future step
    data.update(gt_bboxes_3d=boxes, gt_labels_3d=labels, gt_agent_fut_trajs=trajs, gt_agent_fut_masks=masks)
    lines, line_labels = [], []
    for n in (7, 0):
        base = r(n, 20, 2) * torch.tensor([6.0, 12.0])
        perms = torch.full((n, NUM_PERM, 20, 2), 1e5)
        for i in range(n):
            if i % 2 == 0:   # open poly-line: two point orders, the rest padded as VectorizePloyLine does
                perms[i, 0], perms[i, 1] = base[i], base[i].flip(0)
            else:            # closed polygon: every cyclic shift in both directions
                for s in range(19):
                    perms[i, s] = torch.roll(base[i], s, 0)
                    perms[i, 19 + s] = torch.roll(base[i].flip(0), s, 0)
        lines.append(perms)
        line_labels.append(torch.randint(0, MAP_CLS, (n,), generator=g))
    data.update(gt_map_pts=lines, gt_map_labels=line_labels)
    data["ego_status"] = r(BS, 6)
    data["ego_status_mask"] = (u(BS, 6) > 0.2).float()
    data["gt_ego_fut_cmd"] = torch.tensor([[0, 0, 0, 1, 0, 0], [0, 1, 0, 0, 0, 0]], dtype=torch.float32)
    for key in ("fut_trajs_2hz", "fut_trajs_5hz", "spat_trajs_2m", "spat_trajs_5m"):
        kind, _, rate = key.split("_")
        data[f"gt_ego_{kind}_trajs_{rate}"] = r(BS, TS, 2) * (0.2 if rate == "5hz" else 1.5)
        data[f"gt_ego_{kind}_masks_{rate}"] = (u(BS, TS) > 0.15).float()
    data["gt_ego_fut_trajs_5hz"][1] *= 0.02   # second sample nearly standing still: lowest speed bucket
    return data


def decode_inputs():
    """Head outputs + data of sample 0 for the result-decoder golden case: the loss case's tensors with a dozen
    confident, car-sized agents scattered around the ego's candidate paths so that the plan rescoring has
    collisions to find."""
    outs = [{k: ([t[:1] if t is not None else None for t in v] if isinstance(v, list) else v) for k, v in o.items()}
            for o in head_outputs()]
    det = outs[0]
    g = torch.Generator().manual_seed(4321)
    det["classification"][-1][0] -= 3.0        # background queries stay under the rescoring's confidence threshold
    det["classification"][-1][0, :12] += 9.0   # a dozen confident agents
    radius, angle = 7.0 + 5.0 * torch.rand(12, generator=g), 2 * math.pi * torch.rand(12, generator=g)
    det["prediction"][-1][0, :12, :2] = torch.stack([radius * angle.cos(), radius * angle.sin()], dim=-1)
    det["prediction"][-1][0, :12, 3:6] = torch.log(torch.tensor([2.0, 4.5, 1.6]))
    data = {k: (v[:1] if isinstance(v, torch.Tensor) else v) for k, v in ground_truth().items()}
    return outs, data
