"""The decoder's training objective through the fused loss kernels (hip-ad_amd/csrc/lossprog.hip, include/hipad.h
"The decoder's training objective in a handful of launches").

``FusedObjective`` is driven by ``criterion.DecoderLoss`` (projects/mmdet3d_plugin/models/criterion.py), which keeps the
torch-op formulation of the same arithmetic for CPU tensors, the reference-parity tests with replayed discrete choices
and configurations the kernels do not cover.  Two stages, so that a data-parallel step can exchange the positive counts
in between (hipad_amd.compat.CountExchange):

  assign(outs, gt)     cost matrices -> device Hungarian -> inverse maps + local positive counts   (det, map)
  losses(outs, ...)    every loss term of all decoder layers + d(term)/d(prediction), one launch per task

The autograd node (``_Objective``) returns the vector of the 15 loss terms; its backward is ONE launch that scales the
stored gradients by the upstream gradient of each term.
"""
import ctypes

import torch
from torch.autograd.function import Function, once_differentiable

from . import lib as _lib

MAX_LAYERS, MAX_CLSWISE, MAX_GROUPS, MAX_INTERVALS, MAX_BUCKETS, MAX_SEGMENTS = 8, 4, 16, 4, 8, 16
c_float, c_int, c_void_p = ctypes.c_float, ctypes.c_int, ctypes.c_void_p

TERMS = ("det_loss_cls", "det_loss_box", "det_loss_cns", "det_loss_yns", "map_loss_cls", "map_loss_line",
         "motion_loss_cls", "motion_loss_reg", "plan_loss_temp_cls", "plan_loss_temp_reg", "plan_loss_spat_cls",
         "plan_loss_spat_reg", "plan_loss_speed_cls", "plan_loss_speed_reg", "ego_loss_status")
T_DET, T_MAP, T_MOTION, T_PLAN, T_EGO = 0, 4, 6, 8, 14


class LayerPtrs(ctypes.Structure):
    _fields_ = [("p", c_void_p * MAX_LAYERS)]


class DetCfg(ctypes.Structure):
    _fields_ = [("cost_alpha", c_float), ("cost_gamma", c_float), ("cost_eps", c_float), ("cost_cls_weight", c_float),
                ("cost_box_weight", c_float), ("cost_reg_weights", c_float * 10), ("num_cls_wise", c_int),
                ("cls_wise_label", c_int * MAX_CLSWISE), ("cls_wise_weights", (c_float * 10) * MAX_CLSWISE),
                ("focal_alpha", c_float), ("focal_gamma", c_float), ("w_cls", c_float), ("w_box", c_float),
                ("w_cns", c_float), ("w_yns", c_float), ("gauss_alpha", c_float), ("cls_threshold", c_float),
                ("loss_reg_weights", c_float * 10), ("cns_index", c_int), ("yns_index", c_int)]


class MapCfg(ctypes.Structure):
    _fields_ = [("origin_x", c_float), ("origin_y", c_float), ("norm_x", c_float), ("norm_y", c_float),
                ("cost_cls_weight", c_float), ("cost_reg_weight", c_float), ("cost_beta", c_float),
                ("focal_alpha", c_float), ("focal_gamma", c_float), ("w_cls", c_float), ("w_line", c_float),
                ("loss_beta", c_float), ("cls_threshold", c_float), ("reg_weights", c_float * 40)]


class MotionCfg(ctypes.Structure):
    _fields_ = [("focal_alpha", c_float), ("focal_gamma", c_float), ("w_cls", c_float), ("w_reg", c_float)]


class PlanCfg(ctypes.Structure):
    _fields_ = [("kind", c_int * MAX_GROUPS), ("gt_traj", c_void_p * MAX_GROUPS), ("gt_mask", c_void_p * MAX_GROUPS),
                ("ref_group", c_int), ("num_intervals", c_int), ("interval_size", c_int * MAX_INTERVALS),
                ("interval_group", (c_int * MAX_BUCKETS) * MAX_INTERVALS),
                ("bucket_lo", (c_float * MAX_BUCKETS) * MAX_INTERVALS), ("bucket_hi", (c_float * MAX_BUCKETS) * MAX_INTERVALS),
                ("speed_traj", c_void_p), ("speed_mask", c_void_p), ("speed_interval", c_float),
                ("focal_alpha", c_float), ("focal_gamma", c_float), ("w_cls", c_float), ("w_reg", c_float),
                ("ego_status", c_void_p), ("ego_status_mask", c_void_p), ("w_status", c_float)]


class Segment(ctypes.Structure):
    _fields_ = [("offset", ctypes.c_longlong), ("count", ctypes.c_longlong), ("extra_offset", ctypes.c_longlong),
                ("width", c_int), ("table_offset", c_int), ("extra_cols", c_int), ("extra_term", c_int)]


def _ptrs(tensors):
    t = LayerPtrs()
    for i, x in enumerate(tensors):
        t.p[i] = x.data_ptr()
    return t


def _f32c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


def _call(name, *args):
    _lib.check(getattr(_lib.load(), name)(*args), name)


class _Objective(Function):
    """terms (15,) = the objective's loss terms summed over the decoder layers; inputs = every prediction tensor."""

    @staticmethod
    def forward(ctx, prog, *preds):
        vec, ctx.grads, ctx.plan = prog._run(preds)
        return vec

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        flat, shapes = ctx.grads          # kept: the node may run again (retain_graph), each time into a fresh buffer
        segs, table = ctx.plan
        arr = (Segment * len(segs))(*segs)
        out = torch.empty_like(flat)
        with torch.cuda.device(flat.device):
            _call("hipad_loss_scale", out.data_ptr(), flat.data_ptr(), _f32c(g).data_ptr(), table.data_ptr(),
                  ctypes.addressof(arr), len(segs), _lib.stream_ptr(flat.device))
        views = []
        for off, cnt, shape, n in shapes:
            blk = out[off:off + cnt].view((cnt // max(1, int(torch.Size(shape).numel())),) + tuple(shape))
            views += [blk[l] for l in range(n)]
        return (None,) + tuple(views)


class FusedObjective:
    """Built once per decoder from its samplers / loss modules (``from_decoder``); see the module docstring."""

    def __init__(self, dec):
        import projects.mmdet3d_plugin.core.box3d as B3
        self.tasks = list(dec.task_select)
        self.with_ego = "ego" in self.tasks
        if "det" in self.tasks:
            smp, reg = dec.det_sampler, dec.loss_det_reg
            c = self.det_cfg = DetCfg()
            c.cost_alpha, c.cost_gamma, c.cost_eps = smp.alpha, smp.gamma, smp.eps
            c.cost_cls_weight, c.cost_box_weight = smp.cls_weight, smp.box_weight
            for i, v in enumerate(smp.reg_weights):
                c.cost_reg_weights[i] = v
            cw = smp.cls_wise_reg_weights or {}
            c.num_cls_wise = len(cw)
            for k, (label, ws) in enumerate(cw.items()):
                c.cls_wise_label[k] = int(label)
                for i, v in enumerate(ws):
                    c.cls_wise_weights[k][i] = v
            c.focal_alpha, c.focal_gamma, c.w_cls = dec.loss_det_cls.alpha, dec.loss_det_cls.gamma, dec.loss_det_cls.loss_weight
            c.w_box, c.w_cns, c.w_yns = reg.loss_box.loss_weight, reg.loss_cns.loss_weight, reg.loss_yns.loss_weight
            c.gauss_alpha = reg.loss_yns.alpha
            c.cls_threshold = dec.cls_threshold_to_reg
            for i, v in enumerate(dec.det_reg_weights):
                c.loss_reg_weights[i] = v
            c.cns_index, c.yns_index = B3.CNS, B3.YNS
        if "map" in self.tasks:
            smp, reg = dec.map_sampler, dec.loss_map_reg
            c = self.map_cfg = MapCfg()
            c.origin_x, c.origin_y = -reg.roi_size[0] / 2, -reg.roi_size[1] / 2
            c.norm_x, c.norm_y = reg.roi_size[0] + 1e-5, reg.roi_size[1] + 1e-5
            c.cost_cls_weight, c.cost_reg_weight, c.cost_beta = smp.cls_cost_weight, smp.reg_cost_weight, smp.reg_cost_beta
            c.focal_alpha, c.focal_gamma, c.w_cls = dec.loss_map_cls.alpha, dec.loss_map_cls.gamma, dec.loss_map_cls.loss_weight
            c.w_line, c.loss_beta = reg.loss_line.loss_weight, reg.loss_line.beta
            c.cls_threshold = dec.cls_threshold_to_reg
            for i, v in enumerate(dec.map_reg_weights):
                c.reg_weights[i] = v
        if "motion" in self.tasks:
            c = self.motion_cfg = MotionCfg()
            c.focal_alpha, c.focal_gamma = dec.loss_motion_cls.alpha, dec.loss_motion_cls.gamma
            c.w_cls, c.w_reg = dec.loss_motion_cls.loss_weight, dec.loss_motion_reg.loss_weight
        self._tables = {}

    # ---- eligibility ---------------------------------------------------------------------------------------------
    @staticmethod
    def supports(dec):
        """The kernels cover what the two HiP-AD configs build; anything else stays on the torch-op path."""
        from projects.mmdet3d_plugin.models import criterion as C
        try:
            ok = dec.combine_layer_loss and set(dec.task_select) <= {"det", "map", "motion", "plan", "ego"}
            if "det" in dec.task_select:
                r = dec.loss_det_reg
                ok = ok and isinstance(dec.loss_det_cls, C.FocalLoss) and isinstance(r, C.SparseBox3DLoss) \
                    and isinstance(r.loss_box, C.L1Loss) and isinstance(r.loss_cns, C.CrossEntropyLoss) \
                    and isinstance(r.loss_yns, C.GaussianFocalLoss) and r.loss_yns.gamma == 4.0 \
                    and len(dec.det_reg_weights) == 10 and len(dec.det_sampler.reg_weights) == 10 \
                    and len(dec.det_sampler.cls_wise_reg_weights or {}) <= MAX_CLSWISE
            if "map" in dec.task_select:
                ok = ok and isinstance(dec.loss_map_cls, C.FocalLoss) and isinstance(dec.loss_map_reg, C.SparseLineLoss) \
                    and isinstance(dec.loss_map_reg.loss_line, C.LinesL1Loss) and len(dec.map_reg_weights) == 40 \
                    and dec.loss_map_reg.num_sample == 20 and dec.map_sampler.num_sample == 20 \
                    and tuple(dec.map_sampler.roi_size) == tuple(dec.loss_map_reg.roi_size)
            if "motion" in dec.task_select:
                ok = ok and "det" in dec.task_select and isinstance(dec.loss_motion_cls, C.FocalLoss) \
                    and isinstance(dec.loss_motion_reg, C.L1Loss)
            if "plan" in dec.task_select:
                kinds_ok = all(t[0] in ("temp", "spat", "speed") for t in dec.plan_anchor_types)
                single = dec.ego_fut_cmd == 1 and all(getattr(s, "ego_fut_cmd", 1) == 1 for s in (dec.plan_sampler, dec.align_sampler))
                ok = ok and kinds_ok and single and isinstance(dec.loss_plan_cls, C.FocalLoss) \
                    and isinstance(dec.loss_plan_reg, C.L1Loss) and len(dec.plan_anchor_types) <= MAX_GROUPS \
                    and dec.plan_anchor_types[list(dec.plan_anchor_types).index(dec.plan_anchor_refer)][0] in ("temp", "spat")
            if "ego" in dec.task_select:
                ok = ok and "plan" in dec.task_select and dec.with_supervise_ego_status and isinstance(dec.loss_ego_status, C.L1Loss)
            return bool(ok)
        except (AttributeError, ValueError):
            return False

    # ---- stage 1: target assignment ------------------------------------------------------------------------------
    def assign(self, det_output, map_output, gt):
        """-> dict task -> (matched (L*bs, P) int32, counts (2, L) float32 [, order]); also the det index (L*bs, G)."""
        res = {}
        if "det" in self.tasks:
            cls = [_f32c(t.detach()) for t in det_output["classification"]]
            box = [_f32c(t.detach()) for t in det_output["prediction"]]
            g = gt["det"]
            L, (bs, P, C), D, G = len(cls), cls[0].shape, box[0].shape[-1], g["boxes"].shape[1]
            dev = cls[0].device
            cost = torch.empty(L * bs, G, P, dtype=torch.float32, device=dev)
            n_rows = torch.empty(L * bs, dtype=torch.int32, device=dev)
            index = torch.empty(L * bs, G, dtype=torch.int32, device=dev)
            matched = torch.empty(L * bs, P, dtype=torch.int32, device=dev)
            counts = torch.empty(2, L, dtype=torch.float32, device=dev)
            boxes, labels, count = _f32c(g["boxes"]), g["labels"].contiguous(), g["count"].to(torch.int32).contiguous()
            with torch.cuda.device(dev):
                _call("hipad_loss_det_assign", cost.data_ptr(), n_rows.data_ptr(), index.data_ptr(), matched.data_ptr(),
                      counts.data_ptr(), ctypes.byref(_ptrs(cls)), ctypes.byref(_ptrs(box)), boxes.data_ptr(),
                      labels.data_ptr(), count.data_ptr(), ctypes.byref(self.det_cfg), L, bs, P, C, D, G, boxes.shape[-1],
                      _lib.stream_ptr(dev))
            res["det"] = dict(matched=matched, counts=counts, index=index, boxes=boxes, labels=labels, count=count)
        if "map" in self.tasks:
            cls = [_f32c(t.detach()) for t in map_output["classification"]]
            pts = [_f32c(t.detach()) for t in map_output["prediction"]]
            g = gt["map"]
            L, (bs, P, C), G, NP = len(cls), cls[0].shape, g["pts"].shape[1], g["pts"].shape[2]
            dev = cls[0].device
            cost = torch.empty(L * bs, G, P, dtype=torch.float32, device=dev)
            perm = torch.empty(L * bs, G, P, dtype=torch.uint8, device=dev)
            n_rows = torch.empty(L * bs, dtype=torch.int32, device=dev)
            index = torch.empty(L * bs, G, dtype=torch.int32, device=dev)
            matched = torch.empty(L * bs, P, dtype=torch.int32, device=dev)
            order = torch.zeros(L * bs, G, dtype=torch.int32, device=dev)
            counts = torch.empty(2, L, dtype=torch.float32, device=dev)
            gpts, labels, count = _f32c(g["pts"]), g["labels"].contiguous(), g["count"].to(torch.int32).contiguous()
            with torch.cuda.device(dev):
                _call("hipad_loss_map_assign", cost.data_ptr(), perm.data_ptr(), n_rows.data_ptr(), index.data_ptr(),
                      matched.data_ptr(), order.data_ptr(), counts.data_ptr(), ctypes.byref(_ptrs(cls)),
                      ctypes.byref(_ptrs(pts)), gpts.data_ptr(), labels.data_ptr(), count.data_ptr(),
                      ctypes.byref(self.map_cfg), L, bs, P, C, pts[0].shape[-1], G, NP, _lib.stream_ptr(dev))
            res["map"] = dict(matched=matched, counts=counts, index=index, order=order, pts=gpts, labels=labels, count=count)
        return res

    # ---- stage 2: loss terms + gradients ---------------------------------------------------------------------------
    def losses(self, outs, data, gt, assigned, num_pos):
        """outs = (det, map, ego, plan, motion) output dicts; num_pos: dict task -> (L,) float tensor (after the
        cross-rank mean).  -> (15,) differentiable vector of loss terms (TERMS order)."""
        det, mp, ego, plan, motion = outs
        preds, layout = [], []          # flat list of prediction tensors + (task, key, count) bookkeeping
        for task, out, keys in (("det", det, ("classification", "prediction", "quality")), ("map", mp, ("classification", "prediction")),
                                ("motion", motion, ("classification", "prediction")), ("plan", plan, ("classification", "prediction")),
                                ("ego", ego, ("status",))):
            if task not in self.tasks:
                continue
            for key in keys:
                ts = out[key]
                if ts is None or ts[0] is None:
                    continue
                layout.append((task, key, len(ts)))
                preds += list(ts)
        self._ctx = (layout, data, gt, assigned, num_pos)
        try:
            return _Objective.apply(self, *preds)
        finally:
            self._ctx = None

    def _run(self, preds):
        layout, data, gt, assigned, num_pos = self._ctx
        groups, i = {}, 0
        for task, key, n in layout:
            groups[(task, key)] = [_f32c(t.detach()) for t in preds[i:i + n]]
            i += n
        any_t = preds[0]
        dev = any_t.device
        L = layout[0][2]
        if L > MAX_LAYERS:
            raise _lib.HipadError(f"fused objective: {L} decoder layers > {MAX_LAYERS}")
        # one flat gradient buffer; a [L, ...] block per prediction kind, per-layer gradients are views of it
        blocks, off = {}, 0
        for (task, key), ts in groups.items():
            n = L * ts[0].numel()
            blocks[(task, key)] = (off, n, ts[0].shape)
            off += n
        cns_off = off
        if ("det", "quality") in groups:
            bs, P = groups[("det", "prediction")][0].shape[:2]
            off += L * bs * P * 3
        flat = torch.empty(off, dtype=torch.float32, device=dev)
        terms = torch.zeros(len(TERMS), L, dtype=torch.float32, device=dev)
        stream = _lib.stream_ptr(dev)

        def gptr(task, key):
            return flat.data_ptr() + 4 * blocks[(task, key)][0]

        def tptr(first):
            return terms.data_ptr() + 4 * first * L

        with torch.cuda.device(dev):
            if "det" in self.tasks:
                a = assigned["det"]
                cls, box = groups[("det", "classification")], groups[("det", "prediction")]
                qt = groups.get(("det", "quality"))
                bs, P, C = cls[0].shape
                np_det = _f32c(num_pos["det"])
                _call("hipad_loss_det", tptr(T_DET), gptr("det", "classification"), gptr("det", "prediction"),
                      flat.data_ptr() + 4 * cns_off, gptr("det", "quality") if qt is not None else None,
                      ctypes.byref(_ptrs(cls)), ctypes.byref(_ptrs(box)), ctypes.byref(_ptrs(qt)) if qt is not None else None,
                      a["matched"].data_ptr(), np_det.data_ptr(), a["boxes"].data_ptr(), a["labels"].data_ptr(),
                      ctypes.byref(self.det_cfg), L, bs, P, C, box[0].shape[-1], qt[0].shape[-1] if qt is not None else 0,
                      a["boxes"].shape[1], a["boxes"].shape[-1], stream)
            if "map" in self.tasks:
                a = assigned["map"]
                cls, pts = groups[("map", "classification")], groups[("map", "prediction")]
                bs, P, C = cls[0].shape
                np_map = _f32c(num_pos["map"])
                _call("hipad_loss_map", tptr(T_MAP), gptr("map", "classification"), gptr("map", "prediction"),
                      ctypes.byref(_ptrs(cls)), ctypes.byref(_ptrs(pts)), a["matched"].data_ptr(), a["order"].data_ptr(),
                      np_map.data_ptr(), a["pts"].data_ptr(), a["labels"].data_ptr(), ctypes.byref(self.map_cfg), L, bs, P, C,
                      pts[0].shape[-1], a["pts"].shape[1], a["pts"].shape[2], stream)
            if "motion" in self.tasks:
                cls, reg = groups[("motion", "classification")], groups[("motion", "prediction")]
                bs, A, M = cls[0].shape
                T = reg[0].shape[-2]
                g = gt["motion"]
                trajs, masks = _f32c(g["trajs"]), _f32c(g["masks"])
                npm = num_pos["motion"]
                if npm.dtype != torch.float32:
                    npm = npm.float()
                _call("hipad_loss_motion", tptr(T_MOTION), gptr("motion", "classification"), gptr("motion", "prediction"),
                      ctypes.byref(_ptrs(cls)), ctypes.byref(_ptrs(reg)), assigned["det"]["matched"].data_ptr(), npm.data_ptr(),
                      int(npm.stride(0)) if npm.dim() else 0, trajs.data_ptr(), masks.data_ptr(), ctypes.byref(self.motion_cfg),
                      L, bs, A, M, T, trajs.shape[1], stream)
            if "plan" in self.tasks:
                cls, reg = groups[("plan", "classification")], groups[("plan", "prediction")]
                st = groups.get(("ego", "status"))
                bs = cls[0].shape[0]
                cfg, keep, NG, M, T = self._plan_cfg(data, reg[0])
                S = st[0].shape[-1] if st is not None else 0
                if st is not None:
                    es, em = _f32c(data["ego_status"]), _f32c(data["ego_status_mask"])
                    keep += [es, em]
                    cfg.ego_status, cfg.ego_status_mask = es.data_ptr(), em.data_ptr()
                _call("hipad_loss_plan", tptr(T_PLAN), gptr("plan", "classification"), gptr("plan", "prediction"),
                      gptr("ego", "status") if st is not None else None, ctypes.byref(_ptrs(cls)), ctypes.byref(_ptrs(reg)),
                      ctypes.byref(_ptrs(st)) if st is not None else None, ctypes.byref(cfg), L, bs, NG, M, T, S, stream)
        vec = terms.sum(dim=1)
        shapes = [blocks[(task, key)] + (n,) for task, key, n in layout]     # (offset, count, per-layer shape, layers)
        return vec, (flat, shapes), self._segments(blocks, cns_off, groups, dev)

    # ---- helpers -------------------------------------------------------------------------------------------------
    def _plan_cfg(self, data, reg0):
        dec = self._dec()
        types = list(dec.plan_anchor_types)
        NG = len(types)
        T = reg0.shape[-2]
        M = reg0.shape[-3] // NG
        cfg, keep = PlanCfg(), []

        def gt_of(kind):
            key = "fut" if kind[0] in ("temp", "speed") else "spat"
            t, m = _f32c(data[f"gt_ego_{key}_trajs_{kind[1]}"]), _f32c(data[f"gt_ego_{key}_masks_{kind[1]}"])
            keep.extend((t, m))
            return t, m

        intervals = {}
        for gi, kind in enumerate(types):
            cfg.kind[gi] = {"temp": 0, "spat": 1, "speed": 2}[kind[0]]
            t, m = gt_of(kind)
            cfg.gt_traj[gi], cfg.gt_mask[gi] = t.data_ptr(), m.data_ptr()
            if kind[0] == "speed":
                intervals.setdefault(kind[1], []).append(gi)
        cfg.ref_group = types.index(dec.plan_anchor_refer)
        cfg.num_intervals = len(intervals)
        if len(intervals) > MAX_INTERVALS or any(len(v) > MAX_BUCKETS for v in intervals.values()):
            raise _lib.HipadError("fused objective: too many speed intervals / buckets")
        for iv, (name, idx) in enumerate(intervals.items()):
            cfg.interval_size[iv] = len(idx)
            for k, gi in enumerate(idx):
                cfg.interval_group[iv][k] = gi
                cfg.bucket_lo[iv][k], cfg.bucket_hi[iv][k] = float(types[gi][2][0]), float(types[gi][2][1])
        if intervals:
            t, m = gt_of(dec.plan_speed_refer)
            cfg.speed_traj, cfg.speed_mask = t.data_ptr(), m.data_ptr()
            cfg.speed_interval = 1 / float(dec.plan_speed_refer[1].split("hz")[0])
        cfg.focal_alpha, cfg.focal_gamma = dec.loss_plan_cls.alpha, dec.loss_plan_cls.gamma
        cfg.w_cls, cfg.w_reg = dec.loss_plan_cls.loss_weight, dec.loss_plan_reg.loss_weight
        cfg.w_status = dec.loss_ego_status.loss_weight if self.with_ego else 0.0
        return cfg, keep, NG, M, T

    def _segments(self, blocks, cns_off, groups, dev):
        """Segment descriptors + the column -> term table of the backward scaling (hipad_loss_scale)."""
        dec = self._dec()
        cols, segs = [], []
        for (task, key), (off, cnt, shape) in blocks.items():
            width = int(shape[-1]) if key != "prediction" or task in ("det", "map") else int(torch.Size(shape[2:]).numel())
            if task == "plan" and key == "classification":
                width = int(shape[-1])
            if task == "det":
                table = {"classification": [T_DET] * width, "prediction": [T_DET + 1] * width,
                         "quality": [T_DET + 2 if q == self.det_cfg.cns_index else T_DET + 3 for q in range(width)]}[key]
            elif task == "map":
                table = [T_MAP + (0 if key == "classification" else 1)] * width
            elif task == "motion":
                table = [T_MOTION + (0 if key == "classification" else 1)] * width
            elif task == "plan":
                NG = len(dec.plan_anchor_types)
                per = width // NG
                kinds = [{"temp": 0, "spat": 1, "speed": 2}[t[0]] for t in dec.plan_anchor_types]
                table = [T_PLAN + 2 * kinds[c // per] + (0 if key == "classification" else 1) for c in range(width)]
            else:
                table = [T_EGO] * width
            seg = Segment(off, cnt, -1, width, len(cols), 0, 0)
            if task == "det" and key == "prediction" and ("det", "quality") in groups:
                seg.extra_offset, seg.extra_cols, seg.extra_term = cns_off, 3, T_DET + 2
            cols += table
            segs.append(seg)
        key = (tuple(cols), dev)
        table = self._tables.get(key)
        if table is None:
            table = self._tables[key] = torch.tensor(cols, dtype=torch.int8, device=dev)
        return segs, table

    def _dec(self):
        return self._decoder()

    def bind(self, dec):
        import weakref
        self._decoder = weakref.ref(dec)
        return self
