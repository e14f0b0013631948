"""Generate the golden fixtures in tests/golden/*.npz by RUNNING THE REFERENCE'S OWN PYTHON.

BUILD CONTAINER ONLY (needs /root/reference; the GPU box has neither it nor this need:
the fixtures are committed).  Usage:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--only NAME]

What runs here is the reference's code, imported from where it lies through
tests/golden/_ref_shim.py (third-party names only are stubbed):
  * ops.feature_maps_format                      (ops/__init__.py:33-103)
  * DeformableFeatureAggregation.project_points  (models/blocks.py:216-225)
  * ...feature_sampling / multi_view_level_fusion (models/blocks.py:227-264), the
    reference's CPU fallback for the CUDA op, with the CUDA kernel's border rule
    (sample kept iff 0<x<1 and 0<y<1, deformable_aggregation_cuda.cu:168-171) applied as
    a mask -- SURVEY.md section 8c's oracle definition; gradients from torch autograd
  * ..._get_weights, forward                      (models/blocks.py:124-214)
  * SparseBox3DKeyPointsGenerator (models/det/blocks.py:159-224),
    SparsePoint3DKeyPointsGenerator (models/map/blocks.py:137-225)
  * --only pipeline: ResizeCropFlipImage (datasets/pipelines/augment.py:11-94), Bench2DriveDataset.get_augmentation
    (datasets/bench2drive_dataset.py:709-751), GroupInBatchSampler (datasets/samplers/group_in_batch_sampler.py:48-178),
    BBoxRotation (augment.py:95-138), NuScenesSparse4DAdaptor (pipelines/transform.py:107-168)
Only data (inputs, parameters drawn from a seed, outputs) is stored.
"""
import argparse
import os
import re
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _ref_shim as S  # noqa: E402

S.install()
ref_ops = S.ref_import("ops")
ref_blocks = S.ref_import("models.blocks")
ref_det = S.ref_import("models.det.blocks")
ref_map = S.ref_import("models.map.blocks")
DFA = ref_blocks.DeformableFeatureAggregation

import hipad_amd.synthetic as syn  # noqa: E402


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)  keys={sorted(out)}")


# ----------------------------------------------------------------------------------------
# the reference fallback, driven with explicit 2-D locations
# ----------------------------------------------------------------------------------------
def ref_daf(col_feats, spatial_shape, scale_start_index, loc, weights, num_groups):
    """Reference CPU path on the formatted triple.

    loc (bs,A,P,cams,2), weights (bs,A,P,cams,L,G) -- the CUDA op's argument layout
    (ops/src/deformable_aggregation.cpp:23-29).  Returns out (bs,A,C).
    """
    bs, A, P, cams = loc.shape[:4]
    # formatted triple -> per-level (bs,cams,C,h,w) maps, by the reference's own inverse
    groups = ref_ops.feature_maps_format([col_feats, spatial_shape, scale_start_index], inverse=True)
    assert len(groups) == 1, "fixtures use one camera group"
    level_maps = groups[0]
    L = len(level_maps)
    C = col_feats.shape[-1]

    # feature_sampling() calls DeformableFeatureAggregation.project_points(key_points, M, wh)
    # and expects (bs,cams,A,P,2); hand it our locations instead of projecting.
    loc_cam_major = loc.permute(0, 3, 1, 2, 4)
    orig = DFA.project_points
    DFA.project_points = staticmethod(lambda kp, pm, wh=None: loc_cam_major)
    try:
        dummy_kp = loc.new_zeros(bs, A, P, 3)
        feats = DFA.feature_sampling(level_maps, dummy_kp, None, None)  # (bs,A,cams,L,P,C)
    finally:
        DFA.project_points = staticmethod(orig)
    # CUDA border rule as a mask over (bs,A,P,cam)
    m = ((loc[..., 0] > 0) & (loc[..., 0] < 1) & (loc[..., 1] > 0) & (loc[..., 1] < 1)).to(feats.dtype)
    feats = feats * m.permute(0, 1, 3, 2)[:, :, :, None, :, None]
    holder = type("H", (), {})()
    holder.num_groups, holder.group_dims = num_groups, C // num_groups
    holder.num_pts, holder.embed_dims = P, C
    w_ref = weights.permute(0, 1, 3, 4, 2, 5)  # (bs,A,cams,L,P,G)
    fused = DFA.multi_view_level_fusion(holder, feats, w_ref)  # (bs,A,P,C)
    return fused.sum(dim=2)


def tables(shapes, cams):
    ss = torch.tensor([shapes] * cams, dtype=torch.int64)
    sizes = (ss[..., 0] * ss[..., 1]).flatten()
    start = torch.cat([torch.zeros(1, dtype=torch.int64), sizes.cumsum(0)[:-1]]).reshape(cams, -1)
    return ss, start, int(sizes.sum())


def special_locs(loc, shapes):
    """Plant the border cases the kernel and grid_sample disagree on / edge cases."""
    flat = loc.view(-1, 2)
    h, w = shapes[0]
    cases = [
        (0.0, 0.5), (1.0, 0.5), (0.5, 0.0), (0.5, 1.0),            # exactly on the rim: dropped
        (-0.25 / w, 0.5), (1 + 0.25 / w, 0.5),                    # the sliver grid_sample keeps
        (0.5, -0.25 / h), (0.5, 1 + 0.25 / h),
        (0.25 / w, 0.25 / h), (1 - 0.25 / w, 1 - 0.25 / h),       # inside, corners fall off the map
        (0.5 / w, 0.5 / h), ((w - 0.5) / w, (h - 0.5) / h),       # pixel centres of the rim pixels
        (1e-7, 0.3), (0.3, 1 - 1e-7),
        (-3.0, 0.4), (0.4, 7.5), (1e6, -1e6),
    ]
    for i, (x, y) in enumerate(cases):
        flat[i * 3 % flat.shape[0]] = torch.tensor([x, y])
    return loc


def gen_daf_case(name, shapes, cams, bs, A, P, C, G, seed, lo=-0.2, hi=1.2):
    g = torch.Generator().manual_seed(seed)
    ss, start, F = tables(shapes, cams)
    feat = torch.randn(bs, F, C, generator=g)
    loc = torch.rand(bs, A, P, cams, 2, generator=g) * (hi - lo) + lo
    loc = special_locs(loc, shapes)
    w = torch.softmax(torch.randn(bs, A, P * cams * len(shapes), G, generator=g), dim=2)
    w = w.reshape(bs, A, P, cams, len(shapes), G).contiguous()
    gout = torch.randn(bs, A, C, generator=g)
    feat.requires_grad_(True)
    loc.requires_grad_(True)
    w.requires_grad_(True)
    out = ref_daf(feat, ss, start, loc, w, G)
    out.backward(gout)
    save(name, feat=feat, spatial_shape=ss, scale_start_index=start, loc=loc, weights=w,
         out=out, grad_out=gout, grad_feat=feat.grad, grad_loc=loc.grad, grad_weights=w.grad)


def gen_daf():
    # BASELINE.json configs[0]: 1 cam, 1 level, 32x32 feat, 16 queries x 4 points
    gen_daf_case("daf_unit", [(32, 32)], cams=1, bs=1, A=16, P=4, C=256, G=8, seed=0)
    # 6 cams x 4 levels miniature pyramid (exercises scale_start_index / spatial_shape), bs=2
    gen_daf_case("daf_multicam", [(4, 11), (2, 6), (1, 3), (1, 2)], cams=6, bs=2, A=12, P=5, C=256, G=8, seed=1)
    # odd channel/group split and ragged sizes (generic code path)
    gen_daf_case("daf_ragged", [(5, 7), (3, 4)], cams=3, bs=1, A=7, P=3, C=96, G=4, seed=2)


# ----------------------------------------------------------------------------------------
def gen_format():
    g = torch.Generator().manual_seed(3)
    out = {}
    # data case: small pyramid, 2 samples, 6 cams, 8 channels
    shapes = [(6, 10), (3, 5), (2, 3), (1, 2)]
    maps = [torch.randn(2, 6, 8, h, w, generator=g) for h, w in shapes]
    col, ss, start = ref_ops.feature_maps_format(maps)
    back = ref_ops.feature_maps_format([col, ss, start], inverse=True)
    assert len(back) == 1 and all(torch.equal(a, b) for a, b in zip(back[0], maps))
    for i, m in enumerate(maps):
        out[f"small_map{i}"] = m
    out.update(small_col=col, small_spatial_shape=ss, small_scale_start_index=start)
    # index tables at the two real input sizes (data-free: channel dim 1, zeros)
    for tag, hw in (("704x256", (256, 704)), ("640x352", (352, 640))):
        shp = syn.pyramid_shapes(hw)
        maps = [torch.zeros(1, 6, 1, h, w) for h, w in shp]
        col, ss, start = ref_ops.feature_maps_format(maps)
        out[f"{tag}_spatial_shape"] = ss
        out[f"{tag}_scale_start_index"] = start
        out[f"{tag}_num_feat"] = np.array(col.shape[1])
    save("feature_maps_format", **out)


# ----------------------------------------------------------------------------------------
def ref_lidar2img_from_text():
    """Read the numeric LIDAR2IMG constants from the agent file's text (data, not code)."""
    txt = open(os.path.join(S.REF, "bench2drive/leaderboard/team_code/hipad_b2d_agent.py")).read()
    body = txt[txt.index("LIDAR2IMG = {"): txt.index("LIDAR2CAM = {")]
    nums = [float(x) for x in re.findall(r"[-+]?\d+\.\d+e[-+]\d+", body)]
    return np.array(nums).reshape(6, 4, 4)


def gen_project():
    ref_l2i = ref_lidar2img_from_text()
    mine = syn.bench2drive_lidar2img()
    assert np.allclose(mine, ref_l2i, rtol=1e-6, atol=1e-6), np.abs(mine - ref_l2i).max()
    out = {}
    g = torch.Generator().manual_seed(4)
    for tag, hw in (("704x256", (256, 704)), ("640x352", (352, 640))):
        pm, wh = syn.projection_mats(hw, bs=2)
        pm, wh = torch.from_numpy(pm), torch.from_numpy(wh)
        # points all around the car, incl. behind each camera (clamp path) and on the axis
        kp = (torch.rand(2, 40, 7, 3, generator=g) - 0.5) * torch.tensor([60.0, 120.0, 8.0])
        kp[0, 0, 0] = torch.tensor([0.0, 1.19, -0.24])  # exactly at the front camera centre
        kp[0, 1, 0] = torch.tensor([0.0, 0.0, 0.0])
        p2d = DFA.project_points(kp, pm, wh)  # (bs,cams,A,P,2)
        out[f"{tag}_projection_mat"] = pm
        out[f"{tag}_image_wh"] = wh
        out[f"{tag}_key_points"] = kp
        out[f"{tag}_points_2d"] = p2d
    save("project_points", **out)


# ----------------------------------------------------------------------------------------
def state(mod):
    return {k: v.detach().clone() for k, v in mod.state_dict().items()}


DET_FIX_SCALE = [[0, 0, 0], [0.45, 0, 0], [-0.45, 0, 0], [0, 0.45, 0], [0, -0.45, 0], [0, 0, 0.45], [0, 0, -0.45]]


def cfg_text():
    return open(os.path.join(S.REF, "projects/configs/hipad_b2d_stage2.py")).read()


def dfa_cfgs():
    """The four `*_deformable` config dicts of hipad_b2d_stage2.py, evaluated from its text."""
    txt = cfg_text().replace('"/opt/data/private/project/HiP-AD"', repr(S.REF))
    ns = {}
    exec(compile(txt, "hipad_b2d_stage2.py", "exec"), ns)
    head = ns["model"]["head"] if "head" in ns["model"] else None
    od = None
    for k, v in ns["model"].items():
        if isinstance(v, dict) and "onedecoder_head" in v:
            od = v["onedecoder_head"]
    if od is None:
        od = head["onedecoder_head"]
    return {k: od[f"{k}_deformable"] for k in ("det", "map", "plan", "ego")}, od, ns


def gen_keypoints_and_dfa():
    import copy
    from seeded import seeded, checksum, fill_parameters
    cfgs, od, ns = dfa_cfgs()
    hw = (256, 704)
    shapes = syn.pyramid_shapes(hw)
    # a coarse pyramid keeps the run small: same 6 cams / 4 levels, 1/4 resolution
    shapes_small = [(max(1, h // 4), max(1, w // 4)) for h, w in shapes]
    pm, wh = syn.projection_mats(hw, bs=1)
    pm, wh = torch.from_numpy(pm), torch.from_numpy(wh)
    det_anchor = torch.from_numpy(np.load(os.path.join(S.REF, "data/kmeans/b2d_det_900.npy"))).float()
    map_anchor = torch.from_numpy(np.load(os.path.join(S.REF, "data/kmeans/b2d_map_100.npy"))).float()
    plan_anchor = torch.from_numpy(np.load(os.path.join(S.REF, "data/kmeans/b2d_plan_spat_6x8_2m.npy"))).float()
    anchors = {
        "det": det_anchor[None, ::23][:, :24],                          # (1,24,11)
        "map": map_anchor.reshape(100, -1)[None, ::9][:, :4],           # (1,4,40)
        "plan": plan_anchor.reshape(48, -1)[None, ::7][:, :6],          # (1,6,12)
        "ego": torch.tensor([[[0, 0.5, -1.84 + 0.78, np.log(1.9), np.log(4.9), np.log(1.6), 1, 0, 0, 0, 0]]]).float(),
    }
    maps = [seeded((1, 6, 256, h, w), 700 + i) for i, (h, w) in enumerate(shapes_small)]
    col, ss, start = ref_ops.feature_maps_format(maps)
    out = dict(level_shapes=np.array(shapes_small), col_feats_checksum=checksum(col), spatial_shape=ss,
               scale_start_index=start, projection_mat=pm, image_wh=wh)

    def patched_daf(col_feats, spatial_shape, scale_start_index, points_2d, weights):
        return ref_daf(col_feats, spatial_shape, scale_start_index, points_2d, weights, weights.shape[-1])

    ref_blocks.DAF = patched_daf
    for mi, (name, cfg) in enumerate(cfgs.items()):
        cfg = copy.deepcopy(cfg)
        out[f"{name}_cfg_repr"] = np.array(repr(cfg))
        cfg.pop("type")
        mod = DFA(**cfg)
        # the reference zero-inits weights_fc (uniform softmax); draw seeded params instead so
        # the fixture exercises the whole expression
        out[f"{name}_param_checksum"] = fill_parameters(mod, 1000 * (mi + 1))
        mod.eval()
        anchor = anchors[name]
        A = anchor.shape[1]
        inst = seeded((1, A, 256), 9000 + mi)
        emb = seeded((1, A, 256), 9100 + mi)
        metas = {"projection_mat": pm, "image_wh": wh}
        with torch.no_grad():
            kps = mod.kps_generator(anchor, emb, inst)
            wts = mod._get_weights(inst, emb, metas)
            y = mod(inst, anchor, emb, [col, ss, start], metas)
        print(name, "num_pts", mod.num_pts, "kps", tuple(kps.shape), "weights", tuple(wts.shape), "out", tuple(y.shape))
        out[f"{name}_anchor"] = anchor
        out[f"{name}_key_points"] = kps
        out[f"{name}_weights"] = wts
        out[f"{name}_output"] = y
        out[f"{name}_param_names"] = np.array([k for k, p in mod.named_parameters() if p.requires_grad])
    save("dfa_modules", **out)


def gen_decoder():
    """Whole SparseOneDecoder, reference class, stage-2 config, 2 temporal frames on a small pyramid.

    The two substitutions SURVEY.md section 8c defines for a CPU run of the reference decoder:
    blocks.DAF <- the reference's own torch gather with the CUDA border mask (ref_daf above) and
    FlashAttention.forward <- fp32 softmax(QK^T/sqrt(D))V (flash-attn is not installed)."""
    import copy
    import math
    from types import SimpleNamespace
    from seeded import seeded, checksum, fill_parameters_by_name
    ref_attn = S.ref_import("models.attention")
    S.ref_import("models.separate_attn")
    S.ref_import("models.instance_bank")
    S.ref_import("models.plan.instance_bank"); S.ref_import("models.ego.instance_bank")
    S.ref_import("models.plan.blocks"); S.ref_import("models.ego.blocks"); S.ref_import("models.motion.blocks")
    ref_dec = S.ref_import("models.sparse_onedecoder")

    def sdpa_forward(self, q, kv, causal=False, key_padding_mask=None):
        q, kv = q.float(), kv.float()
        k, v = kv[:, :, 0], kv[:, :, 1]
        att = torch.einsum("bqhd,bkhd->bhqk", q, k) / math.sqrt(q.shape[-1])
        return torch.einsum("bhqk,bkhd->bqhd", att.softmax(-1), v), None

    ref_attn.FlashAttention.forward = sdpa_forward

    def patched_daf(col_feats, spatial_shape, scale_start_index, points_2d, weights):
        return ref_daf(col_feats, spatial_shape, scale_start_index, points_2d, weights, weights.shape[-1])

    ref_blocks.DAF = patched_daf
    hw = (128, 352)  # quarter-size input keeps the CPU run and the fixture small
    _, od, ns = dfa_cfgs()
    txt = cfg_text().replace('"/opt/data/private/project/HiP-AD"', repr(S.REF)).replace(
        "input_shape = (640, 352)", f"input_shape = ({hw[1]}, {hw[0]})")
    ns = {}
    exec(compile(txt, "hipad_b2d_stage2.py", "exec"), ns)
    od = copy.deepcopy(ns["model"]["head"]["onedecoder_head"])
    od.pop("type")
    for k in list(od):
        if k.startswith("loss_") or k.endswith("_sampler") or (k.endswith("_decoder") and k != "num_single_frame_decoder"):
            od[k] = None
    dec = ref_dec.SparseOneDecoder(**od)
    dec.det_sampler = dec.map_sampler = SimpleNamespace(dn_metas=None)
    dec.det_decoder = SimpleNamespace(score_threshold=None)
    dec.init_weights()
    psum = fill_parameters_by_name(dec, 4242)
    dec.eval()
    shapes = syn.pyramid_shapes(hw)
    pm, wh = syn.projection_mats(hw, bs=1)
    out = dict(input_hw=np.array(hw), param_checksum=psum,
               state_keys=np.array(list(dec.state_dict().keys())),
               state_shapes=np.array([str(tuple(v.shape)) for v in dec.state_dict().values()]))
    for step in range(2):
        maps = [seeded((1, 6, 256, h, w), 800 + 10 * step + i, 0.5) for i, (h, w) in enumerate(shapes)]
        fm = ref_ops.feature_maps_format(maps)
        T = syn.ego_motion(step)
        metas = dict(projection_mat=torch.from_numpy(pm), image_wh=torch.from_numpy(wh),
                     timestamp=torch.tensor([0.5 * step], dtype=torch.float64),
                     img_metas=[dict(T_global=T, T_global_inv=np.linalg.inv(T))],
                     gt_ego_fut_cmd=torch.tensor([[0, 0, 0, 1, 0, 0]], dtype=torch.float32),
                     target_point=torch.tensor([[3.0, 25.0]]))
        with torch.no_grad():
            det, mp, ego, plan, motion, _ = dec(None, fm, metas)
        print("step", step, "tokens", dec.total_num_anchor, "temp", dec.total_num_temp_anchor)
        for li in (0, 5):
            out[f"s{step}_det_cls_{li}"] = det["classification"][li]
            out[f"s{step}_det_box_{li}"] = det["prediction"][li]
            out[f"s{step}_det_qt_{li}"] = det["quality"][li]
            out[f"s{step}_map_cls_{li}"] = mp["classification"][li]
            out[f"s{step}_map_pts_{li}"] = mp["prediction"][li]
            out[f"s{step}_plan_cls_{li}"] = plan["classification"][li]
            out[f"s{step}_plan_reg_{li}"] = plan["prediction"][li]
            out[f"s{step}_ego_status_{li}"] = ego["status"][li]
            out[f"s{step}_motion_cls_{li}"] = motion["classification"][li]
        out[f"s{step}_motion_reg_5"] = motion["prediction"][5][:, ::9]
        out[f"s{step}_det_feature"] = det["instance_feature"][:, ::9]
        out[f"s{step}_num_temp"] = np.array(dec.total_num_temp_anchor)
    save("decoder_stage2", **out)


def gen_losses():
    """The reference's SparseOneDecoder.loss (sparse_onedecoder.py:1094-1579) with its own samplers and loss
    modules, on seeded head outputs and ragged ground truth (tests/golden/loss_case.py).  The mmdet primitives
    underneath are the restatements of tests/golden/_mmdet_losses.py (mmdet is not installed)."""
    import copy
    import loss_case as LC
    from seeded import checksum
    for m in ("models.base_target", "models.det.target", "models.det.losses", "models.map.target", "models.map.match_cost",
              "models.map.loss", "models.plan.target", "models.motion.target", "models.instance_bank",
              "models.plan.instance_bank", "models.ego.instance_bank", "models.plan.blocks", "models.ego.blocks",
              "models.motion.blocks", "models.attention", "models.separate_attn"):
        S.ref_import(m)
    ref_dec = S.ref_import("models.sparse_onedecoder")
    txt = cfg_text().replace('"/opt/data/private/project/HiP-AD"', repr(S.REF))
    ns = {}
    exec(compile(txt, "hipad_b2d_stage2.py", "exec"), ns)
    od = copy.deepcopy(ns["model"]["head"]["onedecoder_head"])
    dec = object.__new__(ref_dec.SparseOneDecoder)   # only loss() runs: no parameters, no banks
    torch.nn.Module.__init__(dec)
    build = S.build_from_cfg
    for k in ("det", "map", "plan", "align", "motion"):
        setattr(dec, f"{k}_sampler", build(od[f"{k}_sampler"], S.BBOX_SAMPLERS))
    for k in ("loss_det_cls", "loss_det_reg", "loss_map_cls", "loss_map_reg", "loss_ego_status", "loss_plan_cls",
              "loss_plan_reg", "loss_motion_cls", "loss_motion_reg"):
        setattr(dec, k, build(od[k], S.LOSSES))
    dec.task_select = od["task_select"]
    dec.det_reg_weights, dec.map_reg_weights = od["det_reg_weights"], od["map_reg_weights"]
    dec.cls_threshold_to_reg = od["cls_threshold_to_reg"]
    dec.combine_layer_loss = od.get("combine_layer_loss", True)
    dec.with_supervise_ego_status = od["with_supervise_ego_status"]
    dec.plan_anchor_types = ns["plan_anchor_types"]
    dec.plan_anchor_group = len(ns["plan_anchor_types"])
    dec.plan_anchor_refer, dec.plan_speed_refer = od["plan_anchor_refer"], od["plan_speed_refer"]
    dec.ego_fut_ts, dec.ego_fut_cmd, dec.ego_fut_mode = ns["ego_fut_ts"], ns["ego_fut_cmd"], ns["ego_fut_mode"]
    outs = LC.head_outputs(requires_grad=True)
    data = LC.ground_truth()
    losses = dec.loss(*outs, data)
    total = sum(losses.values())
    total.backward()
    out = {k: v.detach() for k, v in losses.items()}
    out["total"] = total.detach()
    out["loss_keys"] = np.array(sorted(losses))
    det, mp, ego, plan, motion, _ = outs
    out["input_checksum"] = checksum(torch.cat([det["classification"][0].flatten(), plan["prediction"][5].flatten()]))
    for name, group, keys in (("det", det, ("classification", "prediction", "quality")), ("map", mp, ("classification", "prediction")),
                              ("ego", ego, ("status",)), ("plan", plan, ("classification", "prediction")),
                              ("motion", motion, ("classification", "prediction"))):
        for key in keys:
            for li in (0, 5):
                g = group[key][li].grad
                out[f"grad_{name}_{key}_{li}_sum"] = checksum(g)
                out[f"grad_{name}_{key}_{li}_head"] = g.flatten()[:: max(1, g.numel() // 4096)][:4096]
    # the matching itself (last layer), for a direct check of the device Hungarian kernel
    pi, ti = dec.det_sampler.indices[0]
    out["det_match_pred_b0"], out["det_match_gt_b0"] = pi, ti
    print({k: float(v) for k, v in losses.items()})
    save("losses_stage2", **out)


def gen_decode():
    """The reference's result decoders (det / map / motion / plan with collision rescoring) through its own
    SparseOneDecoder.post_process (sparse_onedecoder.py:1581-1605) on the seeded head outputs of loss_case.py,
    sample 0 only (the plan rescoring is written for batch 1, as in the closed loop)."""
    import copy
    import loss_case as LC
    for m in ("models.det.decoder", "models.map.decoder", "models.motion.decoder", "models.plan.decoder",
              "models.instance_bank", "models.plan.instance_bank", "models.ego.instance_bank", "models.plan.blocks",
              "models.ego.blocks", "models.motion.blocks", "models.attention", "models.separate_attn",
              "models.base_target", "models.det.target", "models.det.losses", "models.map.target", "models.map.match_cost",
              "models.map.loss", "models.plan.target", "models.motion.target"):
        S.ref_import(m)
    ref_dec = S.ref_import("models.sparse_onedecoder")
    txt = cfg_text().replace('"/opt/data/private/project/HiP-AD"', repr(S.REF))
    ns = {}
    exec(compile(txt, "hipad_b2d_stage2.py", "exec"), ns)
    od = copy.deepcopy(ns["model"]["head"]["onedecoder_head"])
    dec = object.__new__(ref_dec.SparseOneDecoder)
    torch.nn.Module.__init__(dec)
    for k in ("det", "map", "plan", "motion"):
        setattr(dec, f"{k}_decoder", S.build_from_cfg(od[f"{k}_decoder"], S.BBOX_CODERS))
    dec.task_select, dec.with_supervise_ego_status = od["task_select"], od["with_supervise_ego_status"]
    outs, data = LC.decode_inputs()
    det, mp, ego, plan, motion, _ = outs
    with torch.no_grad():
        det_r, map_r, ego_r, plan_r, motion_r = dec.post_process(det, mp, ego, plan, motion, data)
    out = dict(det_boxes=det_r[0]["boxes_3d"], det_scores=det_r[0]["scores_3d"], det_labels=det_r[0]["labels_3d"],
               det_cls_scores=det_r[0]["cls_scores"], map_vectors=np.stack(map_r[0]["vectors"]), map_scores=map_r[0]["scores"],
               map_labels=map_r[0]["labels"], motion_trajs=motion_r[0]["trajs_3d"], motion_scores=motion_r[0]["trajs_score"],
               plan_keys=np.array(sorted(plan_r[0])))
    for k, v in plan_r[0].items():
        out[k] = v
    print({k: tuple(np.asarray(v).shape) for k, v in out.items()})
    save("decode_stage2", **out)


def _stub_dataset_deps():
    """Third-party names the reference's dataset modules import at module level (none of them is used by the functions
    run below): permissive empty modules."""
    import types

    class Any(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return type(k, (), {})

    for n in ["shapely", "shapely.geometry", "prettytable", "nuscenes", "nuscenes.eval", "nuscenes.eval.common",
              "nuscenes.eval.common.utils", "nuscenes.eval.common.data_classes", "nuscenes.eval.detection",
              "nuscenes.eval.detection.data_classes", "nuscenes.utils", "nuscenes.utils.data_classes",
              "nuscenes.eval.detection.constants", "nuscenes.eval.detection.utils", "pyquaternion", "mmcv.fileio",
              "mmcv.fileio.io", "mmdet.datasets", "mmdet.datasets.pipelines", "mmdet.datasets.builder"]:
        if n not in sys.modules:
            sys.modules[n] = Any(n)
    sys.modules["mmcv.utils"].print_log = print
    sys.modules["mmcv.utils"].track_iter_progress = lambda x: x
    sys.modules["mmcv.runner"].get_dist_info = lambda: (0, 1)

    class Reg:
        def register_module(self, *a, **k):
            return lambda c: c

    sys.modules["mmdet.datasets"].DATASETS = Reg()
    sys.modules["mmdet.datasets.builder"].PIPELINES = Reg()
    S._ns("projects.mmdet3d_plugin.datasets.samplers", os.path.join(S.PLUGIN, "datasets", "samplers"))


def gen_pipeline():
    """Image leg + sampler of the reference's data pipeline, run from its own sources:
      * ResizeCropFlipImage.__call__ / _img_transform (datasets/pipelines/augment.py:11-94; PIL does the pixels)
      * Bench2DriveDataset.get_augmentation (datasets/bench2drive_dataset.py:709-751), numpy global RNG seeded
      * GroupInBatchSampler (datasets/samplers/group_in_batch_sampler.py:48-178) on a toy dataset"""
    import types
    _stub_dataset_deps()
    aug_mod = S.ref_import("datasets.pipelines.augment")
    ds_mod = S.ref_import("datasets.bench2drive_dataset")
    smp_mod = S.ref_import("datasets.samplers.group_in_batch_sampler")
    out = {}
    # ---- get_augmentation: 8 training draws + the test-mode value, on a small frame and on the real one ----
    confs = {"small": {"resize_lim": (0.40, 0.47), "final_dim": (28, 64), "bot_pct_lim": (0.0, 0.0), "rot_lim": (-5.4, 5.4),
                       "H": 90, "W": 160, "rand_flip": True, "rot3d_range": [0, 0]},
             "b2d": {"resize_lim": (0.40, 0.47), "final_dim": (256, 704), "bot_pct_lim": (0.0, 0.0), "rot_lim": (-5.4, 5.4),
                     "H": 900, "W": 1600, "rand_flip": True, "rot3d_range": [0, 0]}}
    draws = {}
    for name, conf in confs.items():
        np.random.seed(2024)
        me = types.SimpleNamespace(data_aug_conf=conf, test_mode=False)
        draws[name] = [ds_mod.Bench2DriveDataset.get_augmentation(me) for _ in range(8)]
        me.test_mode = True
        draws[name].append(ds_mod.Bench2DriveDataset.get_augmentation(me))
        out[f"aug_{name}"] = np.array([[d["resize"], *d["resize_dims"], *d["crop"], float(d["flip"]), d["rotate"], d["rotate_3d"]]
                                       for d in draws[name]], np.float64)
    # ---- the image transform on the small frames: 6 cameras, the 9 aug_configs above + two hand-made edge cases ----
    rng = np.random.default_rng(7)
    imgs = rng.integers(0, 256, (6, 90, 160, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:90, 0:160]
    imgs[1] = ((yy[..., None] * np.array([2, 1, 3]) + xx[..., None] * np.array([1, 3, 2])) % 256).astype(np.uint8)   # smooth
    imgs[2, 20:60, 30:120] = 255                                                                                    # edges
    out["src"] = imgs
    cases = list(draws["small"]) + [
        {"resize": 0.5, "crop": (-3, -2, 61, 26), "flip": True, "rotate": -3.3},      # crop box leaves the image: zero fill
        {"resize": 1.0, "crop": (10, 5, 140, 80), "flip": False, "rotate": 5.0},      # no resampling pass at all
    ]
    out["cases"] = np.array([[c["resize"], *c["crop"], float(c["flip"]), c["rotate"]] for c in cases], np.float64)
    tf = aug_mod.ResizeCropFlipImage()
    l2i = rng.normal(size=(6, 4, 4))
    out["lidar2img"] = l2i
    for k, c in enumerate(cases):
        res = dict(img=[im.astype(np.float32) for im in imgs], aug_config=dict(c), lidar2img=[m.copy() for m in l2i])
        res = tf(res)
        out[f"img_{k}"] = np.stack(res["img"]).astype(np.uint8)        # float32 holding integers 0..255: stored as bytes
        assert np.array_equal(out[f"img_{k}"].astype(np.float32), np.stack(res["img"]))
        out[f"lidar2img_{k}"] = np.stack(res["lidar2img"])
    # ---- BBoxRotation (augment.py:95-138): matrices and boxes of one sample ----
    rot = aug_mod.BBoxRotation()
    boxes = rng.normal(size=(7, 9))
    l2g = rng.normal(size=(4, 4))
    out["rot3d_boxes"], out["rot3d_lidar2global"] = boxes, l2g
    for k, ang in enumerate((0.3, -1.1, 0.0)):
        res = rot(dict(aug_config=dict(rotate_3d=ang), lidar2img=[m.copy() for m in l2i], lidar2global=l2g.copy(),
                       gt_bboxes_3d=boxes.copy()))
        out[f"rot3d_{k}_lidar2img"] = np.stack(res["lidar2img"])
        out[f"rot3d_{k}_lidar2global"], out[f"rot3d_{k}_boxes"] = res["lidar2global"], res["gt_bboxes_3d"]
    # ---- NuScenesSparse4DAdaptor (transform.py:107-168) on one sample; DataContainer stand-in keeps the payload ----
    import types as _types

    class DC:  # mmcv.parallel.DataContainer: only the payload matters here
        def __init__(self, data, **kw):
            self.data = data

    sys.modules["mmcv.parallel"] = _types.ModuleType("mmcv.parallel")
    sys.modules["mmcv.parallel"].DataContainer = DC
    sys.modules["mmdet.datasets.pipelines"].to_tensor = lambda x: x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))
    tr_mod = S.ref_import("datasets.pipelines.transform")
    ad_in = dict(lidar2img=[m.copy() for m in l2i], img_shape=[(28, 64, 3)] * 6, lidar2global=l2g.copy() + 3 * np.eye(4),
                 cam_intrinsic=[np.eye(4) * (k + 1) for k in range(6)], instance_inds=np.arange(7),
                 gt_bboxes_3d=np.concatenate([boxes[:, :6], np.linspace(-9, 9, 7)[:, None], boxes[:, 7:]], 1),
                 gt_labels_3d=np.arange(7) % 3, img=[rng.normal(size=(8, 12, 3)).astype(np.float32) for _ in range(2)],
                 gt_ego_fut_cmd=np.eye(3)[1], ego_status=rng.normal(size=10).astype(np.float32),
                 gt_map_labels=np.arange(4), gt_map_pts=rng.normal(size=(4, 3, 20, 2)))
    for k, v in ad_in.items():
        out[f"adaptor_in_{k}"] = np.stack(v) if isinstance(v, list) else np.asarray(v)
    ad_out = tr_mod.NuScenesSparse4DAdaptor()({k: (list(v) if isinstance(v, list) else (v.copy() if hasattr(v, "copy") else v))
                                               for k, v in ad_in.items()})
    for k in ("projection_mat", "image_wh", "T_global_inv", "T_global", "cam_intrinsic", "focal", "instance_id", "gt_bboxes_3d",
              "gt_labels_3d", "img", "gt_ego_fut_cmd", "ego_status", "gt_map_labels", "gt_map_pts"):
        v = ad_out[k]
        v = v.data if isinstance(v, DC) else v
        out[f"adaptor_out_{k}"] = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    # ---- camera matrices of Bench2DriveDataset.get_data_info (bench2drive_dataset.py:763-805) on a made-up record ----
    def pose(seed):
        r = np.random.default_rng(seed)
        q, _ = np.linalg.qr(r.normal(size=(3, 3)))
        m = np.eye(4)
        m[:3, :3], m[:3, 3] = q, r.normal(size=3) * 3
        return m

    sensors = {"LIDAR_TOP": dict(lidar2ego=pose(1), world2lidar=pose(2))}
    for c in range(6):
        sensors[f"CAM_{c}"] = dict(cam2ego=pose(10 + c), data_path=f"v1/cam{c}/00001.jpg",
                                   intrinsic=np.array([[1142.5 + c, 0, 800], [0, 1142.5 + c, 450], [0, 0, 1.0]]))
    sensors["RADAR_FRONT"] = dict(foo=1)
    me = types.SimpleNamespace(data_infos=[dict(folder="town", frame_idx=7, sensors=sensors)], data_root="/data",
                               modality=dict(use_camera=True), get_ann_info=lambda i: {})
    me.invert_pose = lambda p: ds_mod.Bench2DriveDataset.invert_pose(me, p)
    rec = ds_mod.Bench2DriveDataset.get_data_info(me, 0)
    out["record_lidar2ego"], out["record_world2lidar"] = sensors["LIDAR_TOP"]["lidar2ego"], sensors["LIDAR_TOP"]["world2lidar"]
    out["record_cam2ego"] = np.stack([sensors[f"CAM_{c}"]["cam2ego"] for c in range(6)])
    out["record_intrinsic"] = np.stack([sensors[f"CAM_{c}"]["intrinsic"] for c in range(6)])
    for k in ("ego2img", "lidar2img", "lidar2cam", "cam_intrinsic"):
        out[f"record_out_{k}"] = np.stack(rec[k])
    out["record_out_lidar2global"] = rec["lidar2global"]
    out["record_out_img_filename"] = np.array(rec["img_filename"])
    # ---- sampler: 9 sequences of 3..11 frames, two ranks x batch 2, skipping and reversal on ----
    lens = [5, 3, 7, 4, 11, 6, 3, 8, 5]
    flag = np.concatenate([np.full(n, g) for g, n in enumerate(lens)])
    flag = flag[np.random.default_rng(3).permutation(len(flag))]     # a sequence's frames are not contiguous in the dataset
    out["sampler_flag"] = flag

    class Toy:
        keep_consistent_seq_aug = True

        def __init__(self):
            self.flag = flag
            self.n = 0

        def __len__(self):
            return len(self.flag)

        def get_augmentation(self):
            self.n += 1
            return self.n

    for rank in (0, 1):
        for keep in (True, False):
            np.random.seed(100 + rank)
            ds = Toy()
            ds.keep_consistent_seq_aug = keep
            sm = smp_mod.GroupInBatchSampler(ds, batch_size=2, world_size=2, rank=rank, seed=11, skip_prob=0.15,
                                             sequence_flip_prob=0.3)
            it = iter(sm)
            rows = []
            for _ in range(80):
                rows.append([[d["idx"], d["aug_config"]] for d in next(it)])
            out[f"sampler_rank{rank}_keep{int(keep)}"] = np.array(rows, np.int64)
    save("image_pipeline", **out)


GENS = {"pipeline": gen_pipeline, "decode": gen_decode, "losses": gen_losses, "daf": gen_daf, "format": gen_format, "project": gen_project, "dfa": gen_keypoints_and_dfa,
        "decoder": gen_decoder}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None, choices=sorted(GENS))
    a = ap.parse_args()
    torch.set_num_threads(8)
    for k, fn in GENS.items():
        if a.only in (None, k):
            fn()
