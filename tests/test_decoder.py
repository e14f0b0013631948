"""Whole unified decoder against the reference's SparseOneDecoder (tests/golden/decoder_stage2.npz:
reference class, stage-2 config, seeded parameters, two temporal frames; see make_golden.gen_decoder).

CPU: configs build the same model dict, module tree has the reference's state_dict keys and shapes
(checkpoint compatibility).  GPU: numerical parity of every head output on both frames."""
import os

import numpy as np
import pytest
import torch

from seeded import fill_parameters_by_name, seeded

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CFG = "/root/reference/projects/configs/hipad_b2d_stage{}.py"


def build_decoder(hw):
    import projects.mmdet3d_plugin.models  # noqa: F401  (registers everything)
    from hipad_amd.compat import HEADS, Config, build_from_cfg
    from projects.configs._hipad_b2d_common import hipad_b2d
    cfg = hipad_b2d(stage=2, input_shape=(hw[1], hw[0]))
    dec = build_from_cfg(cfg["model"]["head"]["onedecoder_head"], HEADS)
    dec.init_weights()
    return dec


def test_state_dict_matches_reference_decoder(golden):
    z = golden("decoder_stage2")
    dec = build_decoder(tuple(z["input_hw"]))
    mine = {k: str(tuple(v.shape)) for k, v in dec.state_dict().items()}
    ref = dict(zip(z["state_keys"].tolist(), z["state_shapes"].tolist()))
    assert set(mine) == set(ref), (sorted(set(ref) - set(mine))[:8], sorted(set(mine) - set(ref))[:8])
    assert all(mine[k] == ref[k] for k in ref)
    n = sum(p.numel() for p in dec.parameters())
    assert abs(n - 70.81e6) < 0.02e6  # 70.81 M parameters


@pytest.mark.skipif(not os.path.exists(REF_CFG.format(2)), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("stage", [1, 2])
def test_config_model_dict_equals_reference(stage):
    from hipad_amd.compat import Config
    ns = {}
    exec(compile(open(REF_CFG.format(stage)).read(), "ref_cfg", "exec"), ns)
    mine = Config.fromfile(os.path.join(ROOT, f"projects/configs/hipad_b2d_stage{stage}.py"))

    def norm(o, root):
        if isinstance(o, dict):
            return {str(k): norm(v, root) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [norm(v, root) for v in o]
        return o.replace(root, "<P>") if isinstance(o, str) else o

    assert norm(ns["model"], ns["project_dir"]) == norm(dict(mine["model"]), mine["project_dir"])


def run_two_frames(dec, z, device):
    from hipad_amd import synthetic as syn
    from projects.mmdet3d_plugin.ops import feature_maps_format
    hw = tuple(int(v) for v in z["input_hw"])
    shapes = syn.pyramid_shapes(hw)
    pm, wh = syn.projection_mats(hw, bs=1)
    outs = []
    for step in range(2):
        maps = [seeded((1, 6, 256, h, w), 800 + 10 * step + i, 0.5).to(device) for i, (h, w) in enumerate(shapes)]
        fm = feature_maps_format(maps)
        T = syn.ego_motion(step)
        metas = dict(projection_mat=torch.from_numpy(pm).to(device), image_wh=torch.from_numpy(wh).to(device),
                     timestamp=torch.tensor([0.5 * step], dtype=torch.float64, device=device),
                     img_metas=[dict(T_global=T, T_global_inv=np.linalg.inv(T))],
                     gt_ego_fut_cmd=torch.tensor([[0, 0, 0, 1, 0, 0]], dtype=torch.float32, device=device),
                     target_point=torch.tensor([[3.0, 25.0]], device=device))
        with torch.no_grad():
            outs.append(dec(None, fm, metas))
    return outs


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["torch_fp32", "mfma_bf16"])
def test_decoder_two_frames_match_reference(golden, mode):
    """Every head output of both frames, in the fp32 configuration (library fp32 GEMMs; only attention on
    bf16 operands) and in the bf16 configuration (all Linear layers on the bf16-operand MFMA kernel).  Measured agreement is 1e-4..1e-3 (attention runs on bf16
    operands like the reference's flash-attn; tolerance 1e-2 per BASELINE.json).  Two discrete choices
    in the model turn 4th-digit noise into slot changes and are handled explicitly:
      * the temporal det queries are ordered by top-k over confidences -> on frame 1 det/motion rows
        are matched by box centre before comparing (the match must be a permutation);
      * motion mode anchors are picked by argmax over box classes -> motion heads are compared on the
        anchors where both sides picked the same class.
    The motion class head additionally sees sin/cos(2*pi*metres / 10000^(i/128)) of the rotated mode
    end points: phases of hundreds of radians turn the boxes' 1e-4 yaw noise into percent-level changes
    of a few embedding channels.  It is held to mean-relative 1e-2 and max-relative 0.15."""
    z = golden("decoder_stage2")
    dec = build_decoder(tuple(z["input_hw"]))
    got = fill_parameters_by_name(dec, 4242)
    assert torch.allclose(got, torch.from_numpy(z["param_checksum"]), rtol=1e-9), "seeded parameters drifted"
    from hipad_amd import functional as HF
    dec = dec.cuda().eval()
    with HF.linear_mode(mode):
        outs = run_two_frames(dec, z, "cuda")
    assert dec.total_num_anchor == 1481 and dec.total_num_temp_anchor == int(z["s1_num_temp"]) == 1081
    errs = {}
    # fp32 configuration: 1e-2 (measured 1e-4..1e-3).  bf16 configuration: a decoder layer chains ~20
    # bf16-operand GEMMs with LayerNorms in between; with the fixture's random parameters that compounds
    # to 2-5e-2 on the heads (one module alone stays within 1e-2: tests/test_dfa_gpu.py) -- bounded at 8e-2 on
    # frame 0.  Frame 1 feeds frame 0's noisy outputs back through the top-k temporal selection: its last-layer det
    # heads sit at 8-10e-2 in the max norm and move by a few e-2 with ANY 1e-6 change of arithmetic (observed when the
    # LayerNorm after a Linear moved into that Linear's kernel: bit-identical activations, 1e-6 in the normalised
    # output) -- bounded at 0.13 there.
    # Round 2: the chain kernels keep the activations as hi + lo bf16 pairs (only the weights are rounded), which took the
    # frame-0 worst element from 8e-2 to 3.8e-2 -- bound 5e-2.  This whole-decoder comparison measures how a random-init
    # network AMPLIFIES operand rounding; the per-operator 1e-2 pins of the bf16 configuration are the teacher-forced
    # tests below (test_bf16_configuration_is_pinned_op_by_op, ..._loss_terms_track_fp32).
    TOL = TOL_DEEP = 1e-2 if mode == "torch_fp32" else 5e-2
    # frame 1, last layer, bf16 operands: the worst element over 900 instances is heavy-tailed (a reordered fp32 sum in
    # LayerNorm moves it between 0.09 and 0.23 -- one instance whose temporal top-k neighbour changed), so that case is
    # bounded by the 99th percentile of the element errors plus a loose cap on the worst one
    TOL_FRAME1_DEEP = TOL_DEEP if mode == "torch_fp32" else 0.35
    TOL_FRAME1_DEEP_Q99 = 0.08   # = the frame-0 bound on the worst element

    def npy(t):
        return t.detach().float().cpu().numpy()

    mean_errs = {}

    q99 = {}

    def check(name, a, ref, rows=None):
        assert a.shape == ref.shape, (name, a.shape, ref.shape)
        if rows is not None:
            a, ref = a[:, rows], ref[:, rows]
        d = np.abs(a - ref)
        if "motion_" in name:
            mean_errs[name] = float(d.mean() / np.abs(ref).mean())
            assert d.max() / np.abs(ref).max() < (0.15 if mode == "torch_fp32" else 1.0), (name, d.max() / np.abs(ref).max())
        else:
            errs[name] = float(d.max() / max(1e-9, np.abs(ref).max()))
            q99[name] = float(np.quantile(d, 0.99) / max(1e-9, np.abs(ref).max()))

    for step, (det, mp, ego, plan, motion, _) in enumerate(outs):
        for li in (0, 5):
            ref_box = z[f"s{step}_det_box_{li}"]
            my_box = npy(det["prediction"][li])
            rows = None
            if step == 1 and li > 0:   # after the temporal merge: match slots by box centre
                d2 = ((ref_box[0][:, None, :3] - my_box[0][None, :, :3]) ** 2).sum(-1)
                perm = d2.argmin(1)
                if mode == "torch_fp32":
                    assert len(set(perm.tolist())) == len(perm), "slot matching is not a permutation"
                else:
                    # bf16 noise also changes WHICH instances survive the top-k cut-off: compare the
                    # slots that have an unambiguous partner, and require that to be most of them
                    uniq = np.bincount(perm, minlength=len(perm))[perm] == 1
                    rows = uniq & (d2.min(1) < (0.05 * np.abs(ref_box[0][:, :3]).max()) ** 2)
                    assert rows.mean() > 0.85, rows.mean()
            else:
                perm = np.arange(ref_box.shape[1])
            my_cls = npy(det["classification"][li])[:, perm]
            check(f"s{step}_det_cls_{li}", my_cls, z[f"s{step}_det_cls_{li}"], rows=rows)
            check(f"s{step}_det_box_{li}", my_box[:, perm], ref_box, rows=rows)
            check(f"s{step}_det_qt_{li}", npy(det["quality"][li])[:, perm], z[f"s{step}_det_qt_{li}"], rows=rows)
            check(f"s{step}_map_cls_{li}", npy(mp["classification"][li]), z[f"s{step}_map_cls_{li}"])
            check(f"s{step}_map_pts_{li}", npy(mp["prediction"][li]), z[f"s{step}_map_pts_{li}"])
            check(f"s{step}_plan_cls_{li}", npy(plan["classification"][li]), z[f"s{step}_plan_cls_{li}"])
            check(f"s{step}_plan_reg_{li}", npy(plan["prediction"][li]), z[f"s{step}_plan_reg_{li}"])
            check(f"s{step}_ego_status_{li}", npy(ego["status"][li]), z[f"s{step}_ego_status_{li}"])
            same = my_cls.argmax(-1)[0] == z[f"s{step}_det_cls_{li}"].argmax(-1)[0]
            if rows is not None:
                same = same & rows
            assert same.mean() > (0.9 if mode == "torch_fp32" else 0.6), same.mean()
            if step == 1 and li == 0:
                continue  # layer-0 motion of frame 1 reads the merged slots but layer-0 classes: no common order
            check(f"s{step}_motion_cls_{li}", npy(motion["classification"][li])[:, perm], z[f"s{step}_motion_cls_{li}"], rows=same)
        if step == 0:
            check("s0_motion_reg_5", npy(motion["prediction"][5])[:, ::9], z["s0_motion_reg_5"], rows=same[::9])
            check("s0_det_feature", npy(det["instance_feature"])[:, ::9], z["s0_det_feature"])
    print("relative errors:", {k: round(v, 5) for k, v in sorted(errs.items(), key=lambda kv: -kv[1])[:8]})
    bad = {k: v for k, v in errs.items()
           if not v < (TOL if k.endswith("_0") else (TOL_FRAME1_DEEP if k.startswith("s1_") else TOL_DEEP))}
    assert not bad, bad
    if mode != "torch_fp32":
        tail = {k: v for k, v in q99.items() if k.startswith("s1_") and not k.endswith("_0") and not v < TOL_FRAME1_DEEP_Q99}
        assert not tail, tail
    assert all(v < (1e-2 if mode == "torch_fp32" else 2e-1) for v in mean_errs.values()), mean_errs


def _frame_inputs(z, step, device):
    from hipad_amd import synthetic as syn
    from projects.mmdet3d_plugin.ops import feature_maps_format
    hw = tuple(int(v) for v in z["input_hw"])
    shapes = syn.pyramid_shapes(hw)
    pm, wh = syn.projection_mats(hw, bs=1)
    maps = [seeded((1, 6, 256, h, w), 800 + 10 * step + i, 0.5).to(device) for i, (h, w) in enumerate(shapes)]
    fm = feature_maps_format(maps)
    T = syn.ego_motion(step)
    metas = dict(projection_mat=torch.from_numpy(pm).to(device), image_wh=torch.from_numpy(wh).to(device),
                 timestamp=torch.tensor([0.5 * step], dtype=torch.float64, device=device),
                 img_metas=[dict(T_global=T, T_global_inv=np.linalg.inv(T))],
                 gt_ego_fut_cmd=torch.tensor([[0, 0, 0, 1, 0, 0]], dtype=torch.float32, device=device),
                 target_point=torch.tensor([[3.0, 25.0]], device=device))
    return fm, metas


@pytest.mark.gpu
def test_bf16_configuration_is_pinned_op_by_op(golden):
    """The benchmarked configuration (every Linear on bf16 operands: MFMA GEMM / MLP-chain kernels) against the fp32
    configuration -- which test_decoder_two_frames_match_reference pins to the reference's SparseOneDecoder at 1e-3..1e-4
    -- WITHOUT letting errors compound or discrete choices diverge: the fp32 run records the live state after every op
    of the program (both frames); the bf16 run is teacher-forced, i.e. after each op its state is compared and then
    overwritten with the recorded one, so every op sees the fp32 run's inputs, the temporal top-k selections and the bank
    caches included.  BASELINE.json's bf16 class: aggregated features (output of the `deformable` ops) within 1e-2,
    every head output within 2e-2 (relative to the largest reference magnitude of the tensor)."""
    import copy
    from hipad_amd import functional as HF
    z = golden("decoder_stage2")
    dec = build_decoder(tuple(z["input_hw"]))
    fill_parameters_by_name(dec, 4242)
    dec = dec.cuda().eval()
    fresh = copy.deepcopy(dec)

    def record(log):
        def probe(slot, op, state):
            log.append((slot, op, {k: v.detach().clone() for k, v in state.items()}))
        return probe

    def force(log, errors):
        it = iter(log)

        def probe(slot, op, state):
            rslot, rop, ref = next(it)
            assert (rslot, rop) == (slot, op) and set(ref) == set(state), (slot, op, sorted(set(ref) ^ set(state)))
            for k, v in state.items():
                r = ref[k]
                assert r.shape == v.shape, (slot, op, k, r.shape, v.shape)
                if v.dtype.is_floating_point:
                    err = float((v - r).abs().max() / r.abs().max().clamp_min(1e-9))
                    errors.setdefault((op, k.split(".")[-1] if not k.startswith("out.") else k), []).append(err)
                v.copy_(r)   # teacher forcing
        return probe

    logs, snapshots = [], []
    with torch.no_grad(), HF.linear_mode("torch_fp32"):
        for step in range(2):
            snapshots.append(copy.deepcopy(dec))       # decoder + bank state BEFORE this frame
            log = []
            dec._probe = record(log)
            fm, metas = _frame_inputs(z, step, "cuda")
            dec(None, fm, metas)
            dec._probe = None
            logs.append(log)
    errors = {}
    with torch.no_grad(), HF.linear_mode("mfma_bf16"):
        for step in range(2):
            student = snapshots[step]
            student._probe = force(logs[step], errors)
            fm, metas = _frame_inputs(z, step, "cuda")
            student(None, fm, metas)
    worst = {k: max(v) for k, v in errors.items()}
    print("op-by-op worst relative errors:", {f"{k[0]}:{k[1]}": round(v, 5) for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:16]})
    aggregated = {k: v for k, v in worst.items() if k[0] == "deformable" and k[1] == "feature"}
    heads = {k: v for k, v in worst.items() if k[1].startswith("out.")}
    assert aggregated and heads
    top = lambda d: sorted(((round(v, 5), k) for k, v in d.items()), reverse=True)[:6]  # noqa: E731
    assert all(v < 1e-2 for v in aggregated.values()), top(aggregated)
    # measured (round 3, hi + lo activations in every Linear): motion class 1.29e-2, plan class 1.16e-2, every other head
    # < 9e-3; anchor embeddings 9.7e-3, attention / FFN tokens < 4e-3
    assert all(v < 1.6e-2 for v in heads.values()), top(heads)
    rest = {k: v for k, v in worst.items() if k not in aggregated and k not in heads}
    assert all(v < 1.2e-2 for v in rest.values()), top(rest)


@pytest.mark.gpu
def test_bf16_configuration_loss_terms_track_fp32():
    """Every loss term of the whole model (reference SparseOneDecoder.loss with target assignment, criterion.py) computed
    from the bf16 configuration's head outputs against the same term from the fp32 configuration's, two temporal frames,
    stochastic layers off: BASELINE.json holds losses to 1e-2 in the bf16 class.  The Hungarian assignments are a discrete
    choice, and so are the point order of a matched poly-line, the winning trajectory mode and the class-score gate on
    regression positives (a random-init model has near-ties: one reversed map line moved map_loss_line by 127 % in one
    run out of two), and inside the decoder the temporal top-k selections and the class that picks an agent's motion-mode
    anchors: the bf16 run replays the choices the fp32 run made (hipad_amd.compat.discrete_choice), so the terms compare
    the same instances and matched pairs."""
    import copy
    import warnings
    from hipad_amd import functional as HF
    from hipad_amd.frame import SyntheticFrames, build_detector, frame_losses
    warnings.filterwarnings("ignore")
    torch.manual_seed(7)
    model, _ = build_detector(stage=2, plan_queries=480)
    model.eval()
    model.use_grid_mask = False
    frames = SyntheticFrames(seed=5)
    batches = [frames.next() for _ in range(2)]
    from hipad_amd import compat as CR
    identity = CR.discrete_choice[0]
    recorded = []
    terms = {}
    try:
        for mode in ("torch_fp32", "mfma_bf16"):
            m = copy.deepcopy(model)
            if mode == "torch_fp32":
                def choose(tag, choice):
                    recorded.append((tag, choice.clone()))
                    return choice
            else:
                replay = iter(recorded)

                def choose(tag, choice):
                    rtag, rchoice = next(replay)
                    assert rtag == tag and rchoice.shape == choice.shape, (rtag, tag)
                    return rchoice
            CR.discrete_choice[0] = choose
            with torch.no_grad(), HF.linear_mode(mode):
                for step, (img, data) in enumerate(batches):
                    for k, v in frame_losses(m, img, data).items():
                        terms.setdefault((step, k), {})[mode] = float(v)
    finally:
        CR.discrete_choice[0] = identity
    assert len(recorded) >= 4
    rel = {k: abs(v["mfma_bf16"] - v["torch_fp32"]) / max(abs(v["torch_fp32"]), 1e-6) for k, v in terms.items()}
    print("loss terms, relative difference bf16 vs fp32:", {f"{k[0]}:{k[1]}": round(v, 5) for k, v in sorted(rel.items(), key=lambda kv: -kv[1])[:10]})
    assert len(rel) >= 30
    bad = {k: (round(v, 5), terms[k]) for k, v in rel.items() if not v < 1e-2}
    assert not bad, bad
