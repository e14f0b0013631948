"""Training objective of the unified decoder on the device, in static shapes.

What it computes is the reference's ``SparseOneDecoder.loss`` (models/sparse_onedecoder.py:1094-1579) with its
samplers (det/target.py:66-162, map/target.py:38-62 + 105-160, map/match_cost.py, motion/target.py:71-99,
plan/target.py:80-162) and loss modules (det/losses.py:11-93, map/loss.py:10-120, plus mmdet==2.28.2's
FocalLoss / L1Loss / CrossEntropyLoss / GaussianFocalLoss / FocalLossCost, a third-party package that is not
installed here: their published formulas are restated below).

How it differs underneath (same numbers):
  * ground truth arrives PADDED -- ``(bs, G, ...)`` tensors plus a per-sample count -- instead of Python lists of
    ragged tensors, so every tensor of the step has a static shape;
  * the Hungarian matching runs on the device (``hipad_linear_assignment``), not through ``.cpu().numpy()`` +
    SciPy: no host synchronisation, the whole step stays capturable in a hipGraph;
  * the reference selects the matched rows with boolean-mask indexing (data-dependent shapes, a host sync per
    use); every loss here is a masked sum over all rows -- the same value, because all of those losses are
    sums divided by ``num_pos``.
The registered classes keep the reference's names and constructor keywords so its configs build them.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from hipad_amd.compat import BBOX_SAMPLERS, LOSSES, build_from_cfg, discrete, reduce_mean
from projects.mmdet3d_plugin.core.box3d import CNS, COS_YAW, SIN_YAW, YNS

__all__ = ["FocalLoss", "L1Loss", "CrossEntropyLoss", "GaussianFocalLoss", "SparseBox3DLoss", "LinesL1Loss",
           "SparseLineLoss", "SparseBox3DTarget", "SparsePoint3DTarget", "SparsePlanTarget", "AlignPlanTarget",
           "SparseMotionTarget", "DecoderLoss", "pad_ground_truth", "linear_assignment"]

_EPS = float(torch.finfo(torch.float32).eps)  # mmdet.models.losses.utils.weight_reduce_loss


_CONSTS = {}


def _const(values, like):
    """Small constant vector on ``like``'s device, created once per (values, device, dtype): building it from a
    Python list inside a captured step would be a pageable host->device copy, which a capture cannot hold."""
    key = (tuple(float(v) for v in values), like.device, like.dtype)
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.tensor(key[0], dtype=like.dtype, device=like.device)
    return t


def _reduce(loss, weight=None, avg_factor=None, layers=1):
    """mmdet's weight_reduce_loss for reduction='mean': mean of the elements, or sum / (avg_factor + eps).

    ``layers`` > 1: the leading dimension stacks that many decoder layers (layer-major); the reduction is done
    per layer (``avg_factor`` may be a (layers,) tensor) and a (layers,) vector is returned -- the six decoder
    layers are evaluated by ONE pass of every op instead of six."""
    if weight is not None:
        if weight.dim() == loss.dim() - 1:
            weight = weight.unsqueeze(-1)
        loss = loss * weight.to(loss.dtype)
    if layers == 1:
        return loss.mean() if avg_factor is None else (loss.sum() / (avg_factor + _EPS)).reshape(())
    per = loss.reshape(layers, -1)
    return per.mean(dim=1) if avg_factor is None else per.sum(dim=1) / (avg_factor + _EPS)


def linear_assignment(cost_gt_major, count):
    """(bs, G, P) costs, (bs,) int32 valid-row counts -> (bs, G) int64 prediction index per ground-truth row
    (-1 for padding).  Device kernel for CUDA tensors; SciPy for CPU tensors (host-logic tests only)."""
    if cost_gt_major.is_cuda:
        from hipad_amd import lib
        return lib.linear_assignment(cost_gt_major.detach().float().contiguous(), count.to(torch.int32)).long()
    from scipy.optimize import linear_sum_assignment
    out = torch.full(cost_gt_major.shape[:2], -1, dtype=torch.long)
    for b in range(cost_gt_major.shape[0]):
        n = int(count[b])
        if n:
            rows, cols = linear_sum_assignment(cost_gt_major[b, :n].detach().double().numpy())
            out[b, torch.as_tensor(rows)] = torch.as_tensor(cols)
    return out


# ------------------------------------------------------------------------------------------------
# mmdet loss modules (restated formulas; reduction 'mean' only, which is all the configs use)
# ------------------------------------------------------------------------------------------------
@LOSSES.register_module()
class FocalLoss(nn.Module):
    """Sigmoid focal loss on logits (N, C) with integer targets in [0, C], C = background."""

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction="mean", loss_weight=1.0, activated=False):
        super().__init__()
        if not use_sigmoid or activated or reduction != "mean":
            raise NotImplementedError("only the sigmoid / logits / mean variant used by the HiP-AD configs")
        self.gamma, self.alpha, self.loss_weight = gamma, alpha, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, layers=1):
        if pred.is_cuda and pred.dim() == 2 and pred.dtype == torch.float32:
            from hipad_amd import functional as HF  # value + logit gradient in one kernel
            loss = HF.focal_loss(pred, target, weight, avg_factor, layers, self.alpha, self.gamma)
            return self.loss_weight * (loss.reshape(()) if layers == 1 else loss)
        c = pred.shape[-1]
        t = F.one_hot(target, c + 1)[..., :c].to(pred.dtype)
        p = pred.sigmoid()
        pt = (1 - p) * t + p * (1 - t)
        focal = (self.alpha * t + (1 - self.alpha) * (1 - t)) * pt.pow(self.gamma)
        loss = F.binary_cross_entropy_with_logits(pred, t, reduction="none") * focal
        return self.loss_weight * _reduce(loss, weight, avg_factor, layers)


@LOSSES.register_module()
class L1Loss(nn.Module):
    def __init__(self, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.loss_weight = loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, layers=1):
        return self.loss_weight * _reduce((pred - target).abs(), weight, avg_factor, layers)


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    """Binary cross entropy on logits against soft targets (the ``use_sigmoid=True`` form)."""

    def __init__(self, use_sigmoid=False, reduction="mean", loss_weight=1.0, **kwargs):
        super().__init__()
        if not use_sigmoid:
            raise NotImplementedError("only use_sigmoid=True is used by the HiP-AD configs")
        self.loss_weight = loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, layers=1):
        loss = F.binary_cross_entropy_with_logits(pred, target.to(pred.dtype), reduction="none")
        return self.loss_weight * _reduce(loss, weight, avg_factor, layers)


@LOSSES.register_module()
class GaussianFocalLoss(nn.Module):
    def __init__(self, alpha=2.0, gamma=4.0, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.alpha, self.gamma, self.loss_weight = alpha, gamma, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, layers=1):
        eps = 1e-12
        pos = -(pred + eps).log() * (1 - pred).pow(self.alpha) * target.eq(1).to(pred.dtype)
        neg = -(1 - pred + eps).log() * pred.pow(self.alpha) * (1 - target).pow(self.gamma)
        return self.loss_weight * _reduce(pos + neg, weight, avg_factor, layers)


def _smooth_l1(diff_abs, beta):
    if beta > 0:
        return torch.where(diff_abs < beta, 0.5 * diff_abs * diff_abs / beta, diff_abs - 0.5 * beta)
    return diff_abs


@LOSSES.register_module()
class LinesL1Loss(nn.Module):
    def __init__(self, reduction="mean", loss_weight=1.0, beta=0.5):
        super().__init__()
        self.loss_weight, self.beta = loss_weight, beta

    def forward(self, pred, target, weight=None, avg_factor=None, layers=1):
        loss = _reduce(_smooth_l1((pred - target).abs(), self.beta), weight, avg_factor, layers)
        return loss / (pred.shape[-1] // 2) * self.loss_weight


def _normalize_line(line, num_sample, roi_size):
    """(…, num_sample*2) metric poly-line -> roi-normalised (reference map/loss.py:108-120, map/target.py:64-76)."""
    pts = line.reshape(line.shape[:-1] + (num_sample, -1))
    origin = _const([-roi_size[0] / 2, -roi_size[1] / 2], pts)
    norm = _const([roi_size[0] + 1e-5, roi_size[1] + 1e-5], pts)
    return ((pts - origin) / norm).flatten(-2, -1)


@LOSSES.register_module()
class SparseLineLoss(nn.Module):
    def __init__(self, loss_line, num_sample=20, roi_size=(30, 60)):
        super().__init__()
        self.loss_line = build_from_cfg(loss_line, LOSSES)
        self.num_sample, self.roi_size = num_sample, roi_size

    def forward(self, line, line_target, weight=None, avg_factor=None, prefix="", suffix="", layers=1, **kwargs):
        line = _normalize_line(line, self.num_sample, self.roi_size)
        line_target = _normalize_line(line_target, self.num_sample, self.roi_size)
        return {f"{prefix}loss_line{suffix}": self.loss_line(line, line_target, weight=weight, avg_factor=avg_factor,
                                                             layers=layers)}


@LOSSES.register_module()
class SparseBox3DLoss(nn.Module):
    """Box L1 + centerness + yawness.  ``row_mask`` (N,) marks the rows the reference would have kept with
    boolean indexing; it multiplies every term instead."""

    def __init__(self, loss_box, loss_centerness=None, loss_yawness=None, cls_allow_reverse=None):
        super().__init__()
        if cls_allow_reverse is not None:
            raise NotImplementedError("cls_allow_reverse is unused by the HiP-AD configs")
        self.loss_box = build_from_cfg(loss_box, LOSSES)
        self.loss_cns = build_from_cfg(loss_centerness, LOSSES)
        self.loss_yns = build_from_cfg(loss_yawness, LOSSES)

    def forward(self, box, box_target, weight=None, avg_factor=None, prefix="", suffix="", quality=None,
                cls_target=None, row_mask=None, layers=1, **kwargs):
        m = None if row_mask is None else row_mask.to(box.dtype)
        w = weight if m is None else (m[:, None] if weight is None else weight * m[:, None])
        out = {f"{prefix}loss_box{suffix}": self.loss_box(box, box_target, weight=w, avg_factor=avg_factor, layers=layers)}
        if quality is not None:
            cns, yns = quality[..., CNS], quality[..., YNS].sigmoid()
            cns_target = torch.exp(-torch.norm(box_target[..., :3] - box[..., :3], p=2, dim=-1))
            out[f"{prefix}loss_cns{suffix}"] = self.loss_cns(cns, cns_target, weight=m, avg_factor=avg_factor, layers=layers)
            yaw = slice(SIN_YAW, COS_YAW + 1)  # a slice, not an index list: list indices are uploaded at run time
            yns_target = (F.cosine_similarity(box_target[..., yaw], box[..., yaw], dim=-1) > 0).to(box.dtype)
            out[f"{prefix}loss_yns{suffix}"] = self.loss_yns(yns, yns_target, weight=m, avg_factor=avg_factor, layers=layers)
        return out


# ------------------------------------------------------------------------------------------------
# target assignment
# ------------------------------------------------------------------------------------------------
def _focal_cost(logits, labels, alpha=0.25, gamma=2.0, eps=1e-12):
    """(bs, P, C) logits, (bs, G) labels -> (bs, P, G): mmdet FocalLossCost / det/target.py:126-149."""
    p = logits.sigmoid()
    neg = -(1 - p + eps).log() * (1 - alpha) * p.pow(gamma)
    pos = -(p + eps).log() * alpha * (1 - p).pow(gamma)
    idx = labels.clamp(min=0)[:, None, :].expand(-1, logits.shape[1], -1)
    return torch.gather(pos - neg, 2, idx)


def _scatter_rows(num_pred, index, value, fill=0.0):
    """out[b, index[b, g]] = value[b, g] for index >= 0 (rows with index < 0 are dropped); out (bs, num_pred, …)."""
    bs = index.shape[0]
    slot = torch.where(index >= 0, index, torch.full_like(index, num_pred))  # dropped rows land in a spare slot
    out = value.new_full((bs, num_pred + 1) + value.shape[2:], fill)
    view = slot.reshape(slot.shape + (1,) * (value.dim() - 2)).expand_as(value)
    out.scatter_(1, view, value)
    return out[:, :num_pred]


@BBOX_SAMPLERS.register_module()
class SparseBox3DTarget:
    def __init__(self, cls_weight=2.0, alpha=0.25, gamma=2, eps=1e-12, box_weight=0.25, reg_weights=None,
                 cls_wise_reg_weights=None, num_dn_groups=0, dn_noise_scale=0.5, max_dn_gt=32, add_neg_dn=True,
                 num_temp_dn_groups=0):
        if num_dn_groups or num_temp_dn_groups:
            raise NotImplementedError("denoising queries are off in the HiP-AD configs")
        self.cls_weight, self.box_weight, self.alpha, self.gamma, self.eps = cls_weight, box_weight, alpha, gamma, eps
        self.reg_weights = reg_weights if reg_weights is not None else [1.0] * 8 + [0.0] * 2
        self.cls_wise_reg_weights = cls_wise_reg_weights
        self.dn_metas = None
        self.indices = None

    @staticmethod
    def encode_reg_target(boxes):
        """decoded (…, [x,y,z,w,l,h,yaw,v…]) -> (…, [x,y,z,log w,log l,log h,sin,cos,v…]) (det/target.py:49-64)."""
        return torch.cat([boxes[..., :3], boxes[..., 3:6].log(), boxes[..., 6:7].sin(), boxes[..., 6:7].cos(),
                          boxes[..., 7:]], dim=-1)

    def sample(self, cls_pred, box_pred, gt):
        """gt = dict(boxes (bs, G, D) decoded, labels (bs, G), count (bs,)); returns dense class / box targets and
        regression weights for every prediction, and keeps ``self.indices`` (bs, G) for the motion head."""
        bs, num_pred, num_cls = cls_pred.shape
        labels, count = gt["labels"], gt["count"]
        valid = torch.arange(labels.shape[1], device=labels.device)[None] < count[:, None]
        target = self.encode_reg_target(gt["boxes"])[..., : box_pred.shape[-1]]
        weights = torch.logical_not(target.isnan()).to(target.dtype)
        if self.cls_wise_reg_weights is not None:
            for cls, w in self.cls_wise_reg_weights.items():
                weights = torch.where((labels == cls)[..., None], _const(w, weights), weights)
        with torch.no_grad():  # the matching is not differentiated (the reference detaches the cost)
            cls_cost = _focal_cost(cls_pred, labels, self.alpha, self.gamma, self.eps) * self.cls_weight
            box_cost = ((box_pred[:, :, None] - target[:, None]).abs() * weights[:, None]
                        * _const(self.reg_weights, box_pred)).sum(-1) * self.box_weight
            cost = cls_cost + box_cost
            cost = torch.where(torch.isneginf(cost) | torch.isnan(cost), 1e8, cost)
            index = discrete("assign.det", linear_assignment(cost.transpose(1, 2).contiguous(), count))
        index = torch.where(valid, index, torch.full_like(index, -1))
        self.indices = index
        cls_target = _scatter_rows(num_pred, index, labels, fill=num_cls)
        return cls_target, _scatter_rows(num_pred, index, target), _scatter_rows(num_pred, index, weights)


@BBOX_SAMPLERS.register_module()
class SparsePoint3DTarget:
    """Poly-line targets: focal class cost + permutation-invariant smooth-L1 line cost (map/match_cost.py),
    Hungarian matching, targets taken in the best point order of each matched ground-truth line."""

    def __init__(self, assigner=None, num_dn_groups=0, dn_noise_scale=0.5, max_dn_gt=32, add_neg_dn=True,
                 num_temp_dn_groups=0, num_cls=3, num_sample=20, roi_size=(30, 60)):
        if num_dn_groups or num_temp_dn_groups:
            raise NotImplementedError("denoising queries are off in the HiP-AD configs")
        cost = (assigner or {}).get("cost", {})
        cls_cost, reg_cost = cost.get("cls_cost", {}), cost.get("reg_cost", {})
        if cost.get("iou_cost") is not None or not reg_cost.get("permute", False):
            raise NotImplementedError("only MapQueriesCost(FocalLossCost, LinesL1Cost(permute=True)) is implemented")
        self.cls_cost_weight = cls_cost.get("weight", 1.0)
        self.reg_cost_weight, self.reg_cost_beta = reg_cost.get("weight", 1.0), reg_cost.get("beta", 0.0)
        self.num_cls, self.num_sample, self.roi_size = num_cls, num_sample, roi_size
        self.dn_metas = None

    def sample(self, cls_pred, pts_pred, gt):
        """gt = dict(pts (bs, G, num_permute, num_sample*2), labels (bs, G), count (bs,))."""
        bs, num_pred, num_cls = cls_pred.shape
        labels, count, pts = gt["labels"], gt["count"], gt["pts"]
        valid = torch.arange(labels.shape[1], device=labels.device)[None] < count[:, None]
        with torch.no_grad():  # the matching is not differentiated
            pred_n = _normalize_line(pts_pred, self.num_sample, self.roi_size)
            gt_n = _normalize_line(pts, self.num_sample, self.roi_size)
            cls_cost = _focal_cost(cls_pred, labels) * self.cls_cost_weight
            diff = (pred_n[:, :, None, None] - gt_n[:, None]).abs()                 # (bs, P, G, perm, 2*pts)
            dist = _smooth_l1(diff, self.reg_cost_beta).sum(-1) / (pts_pred.shape[-1] // 2)
            reg_cost, perm = dist.min(dim=-1)                                         # (bs, P, G)
            cost = torch.nan_to_num(cls_cost + reg_cost * self.reg_cost_weight)
            index = discrete("assign.map", linear_assignment(cost.transpose(1, 2).contiguous(), count))
            perm = discrete("line_order", perm)
        index = torch.where(valid, index, torch.full_like(index, -1))
        # point order of ground-truth line g as seen from its matched prediction
        best = torch.gather(perm.transpose(1, 2), 2, index.clamp(min=0)[..., None]).squeeze(-1)   # (bs, G)
        ordered = torch.gather(pts, 2, best[..., None, None].expand(-1, -1, 1, pts.shape[-1])).squeeze(2)
        cls_target = _scatter_rows(num_pred, index, labels, fill=num_cls)
        box_target = _scatter_rows(num_pred, index, ordered)
        return cls_target, box_target, _scatter_rows(num_pred, index, torch.ones_like(ordered))


def _closest_mode(reg_pred, reg_target, reg_weight):
    """(bs, N, modes, ts, 2) offsets vs (bs, N, ts, 2): index of the mode whose cumulative trajectory is closest
    (mean masked L2), reference motion/target.py:5-19 = plan/target.py:7-21."""
    dist = torch.linalg.norm(reg_target.cumsum(dim=-2).unsqueeze(2) - reg_pred.cumsum(dim=-2), dim=-1)
    return discrete("mode", (dist * reg_weight.unsqueeze(2)).mean(dim=-1).argmin(dim=-1))


def _take_mode(reg_pred, mode_idx):
    ts, d = reg_pred.shape[-2:]
    return torch.gather(reg_pred, 2, mode_idx[..., None, None, None].expand(-1, -1, 1, ts, d)).squeeze(2)


@BBOX_SAMPLERS.register_module()
class SparseMotionTarget:
    def sample(self, reg_pred, gt, det_index):
        """reg_pred (bs, A, modes, ts, 2); gt = dict(trajs (bs, G, ts, 2), masks (bs, G, ts), count); det_index
        (bs, G) = the box matching of the last decoder layer (sparse_onedecoder.py:1287)."""
        num_anchor = reg_pred.shape[1]
        reg_target = _scatter_rows(num_anchor, det_index, gt["trajs"].to(reg_pred.dtype))
        reg_weight = _scatter_rows(num_anchor, det_index, gt["masks"].to(reg_pred.dtype))
        num_pos = (det_index >= 0).sum().to(reg_pred.dtype)
        mode = _closest_mode(reg_pred, reg_target, reg_weight)
        return mode, reg_weight.any(dim=-1), _take_mode(reg_pred, mode), reg_target, reg_weight, num_pos


class _PlanTargetBase:
    def __init__(self, ego_fut_ts=6, ego_fut_cmd=3, ego_fut_mode=3):
        self.ego_fut_ts, self.ego_fut_cmd, self.ego_fut_mode = ego_fut_ts, ego_fut_cmd, ego_fut_mode

    def _select_command(self, cls_pred, reg_pred, data):
        bs = reg_pred.shape[0]
        cls_pred = cls_pred.reshape(bs, self.ego_fut_cmd, 1, -1)
        reg_pred = reg_pred.reshape(bs, self.ego_fut_cmd, 1, -1, self.ego_fut_ts, 2)
        if self.ego_fut_cmd > 1:
            rows = torch.arange(bs, device=reg_pred.device)
            cmd = data["gt_ego_fut_cmd"].argmax(dim=-1)
            return cls_pred[rows, cmd], reg_pred[rows, cmd]
        return cls_pred[:, 0], reg_pred[:, 0]


@BBOX_SAMPLERS.register_module()
class SparsePlanTarget(_PlanTargetBase):
    def sample(self, cls_pred, reg_pred, gt_reg_target, gt_reg_mask, data):
        gt_reg_target, gt_reg_mask = gt_reg_target.unsqueeze(1), gt_reg_mask.unsqueeze(1)
        cls_pred, reg_pred = self._select_command(cls_pred, reg_pred, data)
        mode = _closest_mode(reg_pred, gt_reg_target, gt_reg_mask)
        return cls_pred, mode, gt_reg_mask.any(dim=-1), _take_mode(reg_pred, mode), gt_reg_target, gt_reg_mask


@BBOX_SAMPLERS.register_module()
class AlignPlanTarget(_PlanTargetBase):
    def sample(self, cls_pred, reg_pred, gt_reg_target, gt_reg_mask, data, ref_target):
        gt_reg_target, gt_reg_mask = gt_reg_target.unsqueeze(1), gt_reg_mask.unsqueeze(1)
        cls_pred, reg_pred = self._select_command(cls_pred, reg_pred, data)
        return cls_pred, ref_target, gt_reg_mask.any(dim=-1), _take_mode(reg_pred, ref_target), gt_reg_target, gt_reg_mask


# ------------------------------------------------------------------------------------------------
# ground truth: ragged lists (the reference's data format) -> padded tensors
# ------------------------------------------------------------------------------------------------
def _pad(items, length, fill, device):
    tail = tuple(items[0].shape[1:]) if len(items) else ()
    out = torch.full((len(items), length) + tail, fill, dtype=items[0].dtype, device=device)
    count = torch.zeros(len(items), dtype=torch.int32)
    for i, t in enumerate(items):
        n = min(len(t), length)
        out[i, :n] = t[:n].to(device)
        count[i] = n
    return out, count.to(device)


def pad_ground_truth(data, device=None, max_det=None, max_map=None):
    """``gt_padded`` entry for ``data``: the reference's ragged lists (``gt_bboxes_3d`` / ``gt_labels_3d`` /
    ``gt_map_pts`` / ``gt_map_labels`` / ``gt_agent_fut_trajs`` / ``gt_agent_fut_masks``) as padded tensors +
    counts.  Host-side helper (runs outside the captured step)."""
    out = {}
    if "gt_bboxes_3d" in data:
        boxes = list(data["gt_bboxes_3d"])
        device = device or boxes[0].device
        g = max_det or max(1, max(len(b) for b in boxes))
        pb, count = _pad(boxes, g, 1.0, device)  # padding boxes have size 1 so log() stays finite
        pl, _ = _pad(list(data["gt_labels_3d"]), g, 0, device)
        out["det"] = dict(boxes=pb, labels=pl, count=count)
        if "gt_agent_fut_trajs" in data:
            pt, _ = _pad(list(data["gt_agent_fut_trajs"]), g, 0.0, device)
            pm, _ = _pad(list(data["gt_agent_fut_masks"]), g, 0.0, device)
            out["motion"] = dict(trajs=pt, masks=pm, count=count)
    if "gt_map_pts" in data:
        pts = [p.flatten(2, 3) if p.dim() == 4 else p for p in data["gt_map_pts"]]
        device = device or pts[0].device
        g = max_map or max(1, max(len(p) for p in pts))
        pp, count = _pad(pts, g, 0.0, device)
        pl, _ = _pad(list(data["gt_map_labels"]), g, 0, device)
        out["map"] = dict(pts=pp, labels=pl, count=count)
    return out


# ------------------------------------------------------------------------------------------------
# the decoder's loss()
# ------------------------------------------------------------------------------------------------
def _stack_layers(tensors):
    """[L x (bs, ...)] -> (L*bs, ...), layer-major."""
    return torch.stack(list(tensors)).flatten(0, 1)


def _tile(t, layers):
    """(bs, ...) ground truth -> (L*bs, ...) matching _stack_layers."""
    return t.repeat((layers,) + (1,) * (t.dim() - 1))


import os as _os

PLAN_LOSS_BATCHED = _os.environ.get("HIPAD_PLAN_LOSS_BATCHED", "1") == "1"   # 0: the group-by-group form
FUSED_LOSS = _os.environ.get("HIPAD_FUSED_LOSS", "1") == "1"                 # 0: the torch-op formulation below on the GPU too


class LossDict(dict):
    """The loss terms by name, as the reference returns them, plus ``total`` = their sum when the producer already has it
    (the fused objective: one reduction of its term vector instead of a chain of scalar additions)."""

    total = None



class DecoderLoss:
    """``loss()`` of SparseOneDecoder (mixed into the class in sparse_onedecoder.py).

    All decoder layers are evaluated together: their outputs are stacked along the batch dimension, the ground
    truth is tiled, every op (costs, the Hungarian kernel, scatters, loss formulas) runs once for the six layers,
    and only the final reductions are per layer (each layer has its own ``num_pos``)."""

    # ---- the objective through the fused loss kernels (hipad_amd/lossprog.py) ---------------------------------------
    def _fused_objective(self, outs, gt):
        """The FusedObjective of this decoder when this call can go through it: CUDA fp32 predictions, a configuration
        the kernels cover, discrete choices not being recorded / replayed by a test (those hooks live in the torch-op
        formulation), ground-truth padding within the kernels' tables."""
        if not FUSED_LOSS:
            return None
        from hipad_amd import compat as _compat
        if _compat.discrete_choice[0] is not _compat._identity_choice and _compat.discrete_scope[0] != "decoder":
            return None
        probe = next((t for o in outs[:5] for v in o.values() if isinstance(v, list) for t in v if isinstance(t, torch.Tensor)), None)
        if probe is None or not probe.is_cuda:
            return None
        fo = self.__dict__.get("_fused_obj")
        if fo is None:
            from hipad_amd import lossprog
            fo = lossprog.FusedObjective(self).bind(self) if lossprog.FusedObjective.supports(self) else False
            self.__dict__["_fused_obj"] = fo
        if not fo:
            return None
        for o in outs[:5]:
            for v in o.values():
                if isinstance(v, list) and any(isinstance(t, torch.Tensor) and t.dtype != torch.float32 for t in v):
                    return None
        layers = len(outs[0]["classification"]) if "det" in self.task_select else len(outs[3]["classification"])
        if layers > 8 or ("det" in gt and gt["det"]["boxes"].shape[1] > 64) or \
                ("map" in gt and (gt["map"]["pts"].shape[1] > 64 or gt["map"]["pts"].shape[2] > 255 or gt["map"]["pts"].shape[-1] != 40)):
            return None
        return fo

    def _fused_counts(self, assigned):
        """The positive counts of the matched tasks through ``reduce_mean``, in the order of the torch-op formulation."""
        num_pos = {}
        if "det" in self.task_select:
            num_pos["det"] = reduce_mean(assigned["det"]["counts"][0])
        if "map" in self.task_select:
            num_pos["map"] = reduce_mean(assigned["map"]["counts"][0])
        if "motion" in self.task_select:
            raw = assigned["det"]["counts"][1]
            num_pos["motion"] = reduce_mean(raw[-1:].expand(raw.shape[0]))   # the LAST layer's matching serves every layer
        return num_pos

    def _loss_fused(self, fo, outs, data, gt):
        from hipad_amd import lossprog
        assigned = self.__dict__.pop("_assigned", None) or fo.assign(outs[0], outs[1], gt)
        if "det" in assigned:
            self.det_sampler.indices = assigned["det"]["index"]
        vec = fo.losses(outs[:5], data, gt, assigned, self._fused_counts(assigned))
        active = {"det": lossprog.TERMS[0:4], "map": lossprog.TERMS[4:6], "motion": lossprog.TERMS[6:8],
                  "plan": lossprog.TERMS[8:14], "ego": lossprog.TERMS[14:15]}
        kinds = {t[0] for t in getattr(self, "plan_anchor_types", [])}
        out = LossDict()
        for task in ("det", "map", "ego", "motion", "plan"):
            if task in self.task_select:
                for name in active[task]:
                    if task == "plan" and name.split("_")[2] not in kinds:
                        continue
                    out[name] = vec[lossprog.TERMS.index(name)]
        out.total = vec.sum()
        return out

    def positive_counts(self, det_output, map_output, ego_output, plan_output, motion_output, scenes_output, data):
        """Phase 1 of a data-parallel step (hipad_amd.compat.CountExchange): the target assignment of the matched tasks
        and their positive counts, which ``reduce_mean`` hands to the exchange buffer -- in the order ``loss`` asks for
        them again.  The assignments are kept and reused by the following ``loss`` call (not recomputed)."""
        gt = data.get("gt_padded") or pad_ground_truth(data)
        self._sampled = {}
        fo = self._fused_objective((det_output, map_output, ego_output, plan_output, motion_output), gt)
        if fo is not None:
            with torch.no_grad():
                self.__dict__["_assigned"] = fo.assign(det_output, map_output, gt)
                self._fused_counts(self.__dict__["_assigned"])
            return
        with torch.no_grad():
            if "det" in self.task_select:
                self._sampled["det_"] = self._sample_targets(det_output, gt["det"], self.det_sampler, self.det_reg_weights)
                reduce_mean(self._sampled["det_"][-1])
            if "map" in self.task_select:
                self._sampled["map_"] = self._sample_targets(map_output, gt["map"], self.map_sampler, self.map_reg_weights)
                reduce_mean(self._sampled["map_"][-1])
            if "motion" in self.task_select:
                layers = len(motion_output["classification"])
                bs = motion_output["prediction"][0].shape[0]
                det_index = _tile(self.det_sampler.indices[-bs:], layers)
                reduce_mean((det_index >= 0).reshape(layers, -1).sum(dim=1).to(torch.float32))

    def _sample_targets(self, outs, gt, sampler, reg_weights):
        layers = len(outs["classification"])
        cls = _stack_layers(outs["classification"]).detach()
        reg = _stack_layers(outs["prediction"])[..., : len(reg_weights)].detach()
        cls_target, reg_target, weights = sampler.sample(cls, reg, {k: _tile(v, layers) for k, v in gt.items()})
        sampler.layers = layers
        matched = torch.logical_not(torch.all(reg_target == 0, dim=-1))
        return cls_target, reg_target, weights, matched, matched.reshape(layers, -1).sum(dim=1).to(torch.float32)

    def loss(self, det_output, map_output, ego_output, plan_output, motion_output, scenes_output, data):
        gt = data.get("gt_padded") or pad_ground_truth(data)
        outs = (det_output, map_output, ego_output, plan_output, motion_output)
        fo = self._fused_objective(outs, gt)
        if fo is not None:
            return self._loss_fused(fo, outs, data, gt)
        self.__dict__.pop("_assigned", None)
        losses = {}
        if "det" in self.task_select:
            losses.update(self.loss_det(det_output, gt["det"]))
        if "map" in self.task_select:
            losses.update(self.loss_map(map_output, gt["map"]))
        if "ego" in self.task_select:
            losses.update(self.loss_ego(ego_output, data))
        if "motion" in self.task_select:
            losses.update(self.loss_motion(motion_output, gt["motion"]))
        if "plan" in self.task_select:
            losses.update(self.loss_plan(plan_output, data))
        return losses

    def _add(self, output, key, per_layer):
        """per_layer: (L,) vector of one loss term."""
        if self.combine_layer_loss:
            output[key] = output.get(key, 0.0) + per_layer.sum()
        else:
            for layer in range(per_layer.shape[0]):
                output[f"{key}_{layer}"] = per_layer[layer]

    def _matched_losses(self, outs, gt, sampler, reg_weights, loss_cls, loss_reg, prefix, reg_keys):
        layers = len(outs["classification"])
        cls = _stack_layers(outs["classification"])
        reg = _stack_layers(outs["prediction"])[..., : len(reg_weights)]
        qt = None if outs["quality"][0] is None else _stack_layers(outs["quality"])
        kept = getattr(self, "_sampled", None)
        if kept and prefix in kept:   # phase 2 of a data-parallel step: the assignment made in positive_counts()
            cls_target, reg_target, weights, matched, count = kept.pop(prefix)
        else:
            cls_target, reg_target, weights, matched, count = self._sample_targets(outs, gt, sampler, reg_weights)
        num_pos = torch.clamp(reduce_mean(count), min=1.0)   # (L,)
        rows = matched
        if self.cls_threshold_to_reg > 0:
            rows = rows & discrete("cls_gate", cls.max(dim=-1).values.sigmoid() > self.cls_threshold_to_reg)
        cls_loss = loss_cls(cls.flatten(end_dim=1), cls_target.flatten(end_dim=1), avg_factor=num_pos, layers=layers)
        weights = (weights * _const(reg_weights, reg)).flatten(end_dim=1)
        reg_target = reg_target.flatten(end_dim=1)
        reg_target = torch.where(reg_target.isnan(), 0.0, reg_target)
        reg_loss = loss_reg(reg.flatten(end_dim=1), reg_target, weight=weights, avg_factor=num_pos, prefix=prefix,
                            quality=None if qt is None else qt.flatten(end_dim=1),
                            cls_target=cls_target.flatten(end_dim=1), row_mask=rows.flatten(), layers=layers)
        output = {}
        self._add(output, f"{prefix}loss_cls", cls_loss.reshape(layers))
        for key in reg_keys:
            self._add(output, f"{prefix}loss_{key}", reg_loss[f"{prefix}loss_{key}"].reshape(layers))
        return output

    def loss_det(self, outs, gt):
        return self._matched_losses(outs, gt, self.det_sampler, self.det_reg_weights, self.loss_det_cls,
                                    self.loss_det_reg, "det_", ("box", "cns", "yns"))

    def loss_map(self, outs, gt):
        def line_loss(line, target, weight=None, row_mask=None, **kw):  # SparseLineLoss has no row selection of its own
            return self.loss_map_reg(line, target, weight=weight * row_mask.to(weight.dtype)[:, None], **kw)
        return self._matched_losses(outs, gt, self.map_sampler, self.map_reg_weights, self.loss_map_cls,
                                    line_loss, "map_", ("line",))

    def loss_ego(self, outs, data):
        if not self.with_supervise_ego_status:
            raise NotImplementedError("ego trajectory supervision is unused by the HiP-AD configs")
        layers = len(outs["status"])
        status = _stack_layers(outs["status"]).squeeze(1)
        loss = self.loss_ego_status(status, _tile(data["ego_status"], layers), weight=_tile(data["ego_status_mask"], layers),
                                    layers=layers)
        output = {}
        self._add(output, "ego_loss_status", torch.nan_to_num(loss).reshape(layers))
        return output

    def loss_motion(self, outs, gt):
        layers = len(outs["classification"])
        cls, reg = _stack_layers(outs["classification"]), _stack_layers(outs["prediction"])
        bs = reg.shape[0] // layers
        # the box matching of the LAST decoder layer serves every motion layer (sparse_onedecoder.py:1287)
        det_index = _tile(self.det_sampler.indices[-bs:], layers)
        cls_target, cls_weight, best, reg_target, reg_weight, _ = self.motion_sampler.sample(
            reg, {k: _tile(v, layers) for k, v in gt.items()}, det_index)
        num_pos = torch.clamp(reduce_mean((det_index >= 0).reshape(layers, -1).sum(dim=1).to(reg.dtype)), min=1.0)
        cls_loss = self.loss_motion_cls(cls.flatten(end_dim=1), cls_target.flatten(end_dim=1),
                                        weight=cls_weight.flatten(end_dim=1), avg_factor=num_pos, layers=layers)
        reg_loss = self.loss_motion_reg(best.flatten(end_dim=1).cumsum(dim=-2), reg_target.flatten(end_dim=1).cumsum(dim=-2),
                                        weight=reg_weight.flatten(end_dim=1).unsqueeze(-1), avg_factor=num_pos, layers=layers)
        output = {}
        self._add(output, "motion_loss_cls", cls_loss.reshape(layers))
        self._add(output, "motion_loss_reg", reg_loss.reshape(layers))
        return output

    # ---- planning ---------------------------------------------------------------------------------
    def _plan_slice(self, cls, reg, kind):
        modes = reg.size(2) // self.plan_anchor_group
        s = self.ego_fut_cmd * modes * self.plan_anchor_types.index(kind)
        return cls[:, :, s:s + self.ego_fut_cmd * modes], reg[:, :, s:s + self.ego_fut_cmd * modes]

    @staticmethod
    def _plan_gt(data, kind):
        key = "fut" if kind[0] in ("temp", "speed") else "spat"
        return data[f"gt_ego_{key}_trajs_{kind[1]}"], data[f"gt_ego_{key}_masks_{kind[1]}"]

    def _plan_terms(self, cls, target, cls_weight, best, gt_target, gt_mask, layers):
        cls_loss = self.loss_plan_cls(cls.flatten(end_dim=1), target.flatten(end_dim=1), weight=cls_weight.flatten(end_dim=1),
                                      layers=layers)
        reg_loss = self.loss_plan_reg(best.cumsum(dim=-2).flatten(end_dim=1), gt_target.cumsum(dim=-2).flatten(end_dim=1),
                                      weight=gt_mask.flatten(end_dim=1).unsqueeze(-1), layers=layers)
        return cls_loss.reshape(layers), reg_loss.reshape(layers)

    def loss_plan(self, outs, data):
        single_cmd = all(getattr(smp, "ego_fut_cmd", 1) == 1 for smp in (self.plan_sampler, self.align_sampler)) \
            and self.ego_fut_cmd == 1
        known = all(t[0] in ("temp", "spat", "speed") for t in self.plan_anchor_types)
        if single_cmd and known and PLAN_LOSS_BATCHED:
            return self._loss_plan_batched(outs, data)
        return self._loss_plan_general(outs, data)

    def _loss_plan_batched(self, outs, data):
        """The planning terms with ALL anchor groups evaluated together (single driving command, as the stage configs
        have it): the reference walks the ten anchor types one by one (sparse_onedecoder.py:1315-1443) -- ten target
        selections, six classification and six regression losses, each a dozen small launches forward and again backward.
        Here the groups are a tensor dimension: one gather of the reference mode's trajectory / logit for every group,
        ONE focal loss over the aligned groups, ONE over the speed buckets, ONE L1 over all trajectories, with the
        (group, layer) pairs stacked where the loss modules expect the layers, so every term is still reduced per
        group and per layer exactly as in the group-by-group form (_loss_plan_general; tests/test_losses.py compares)."""
        layers = len(outs["classification"])
        cls, reg = _stack_layers(outs["classification"]), _stack_layers(outs["prediction"])
        data = {k: (_tile(v, layers) if k.startswith("gt_ego_") else v) for k, v in data.items()}
        types = list(self.plan_anchor_types)
        G = self.plan_anchor_group
        B, ts = cls.shape[0], reg.shape[-2]
        modes = reg.size(2) // G
        cls_g, reg_g = cls.reshape(B, G, modes), reg.reshape(B, G, modes, ts, 2)
        r = types.index(self.plan_anchor_refer)
        ref_gt, ref_mask = self._plan_gt(data, self.plan_anchor_refer)
        # the winning mode of the reference group decides the target mode of every other group
        _, ref_target, ref_weight, _, _, _ = self.plan_sampler.sample(cls_g[:, r:r + 1], reg_g[:, r:r + 1], ref_gt, ref_mask, data)
        best = torch.gather(reg_g, 2, ref_target[:, :, None, None, None].expand(B, G, 1, ts, 2)).squeeze(2)   # (B, G, ts, 2)
        logit = torch.gather(cls_g, 2, ref_target[:, :, None].expand(B, G, 1)).squeeze(2)                       # (B, G)

        def group_major(t):       # (B, n, ...) -> (n * B, ...): rows ordered (group, layer, sample)
            return t.transpose(0, 1).reshape((-1,) + tuple(t.shape[2:]))

        align = [i for i, t in enumerate(types) if t[0] in ("temp", "spat")]
        n_a = len(align)
        a_idx = _const(align, logit).long() if align != list(range(align[0], align[0] + n_a)) else None
        take = (lambda t: t.index_select(1, a_idx)) if a_idx is not None else (lambda t: t[:, align[0]:align[0] + n_a])
        gts = [self._plan_gt(data, types[i]) for i in align]
        a_gt, a_mask = torch.stack([g for g, _ in gts], 1), torch.stack([m for _, m in gts], 1)      # (B, n_a, ts, 2), (B, n_a, ts)
        a_cls = self.loss_plan_cls(group_major(take(cls_g)), ref_target.flatten().repeat(n_a),
                                   weight=ref_weight.flatten().repeat(n_a), layers=n_a * layers).reshape(n_a, layers)
        # speed groups: per interval the buckets' logits / trajectories of the reference mode; class = the speed bucket
        intervals = {}
        for i, t in enumerate(types):
            if t[0] == "speed":
                intervals.setdefault(t[1], []).append(i)
        s_logits, s_best, s_gt, s_mask, s_bucket = [], [], [], [], []
        if intervals:       # (stage 1 has a single temporal group and no speed groups)
            sp_traj, sp_mask = self._plan_gt(data, self.plan_speed_refer)
            interval = 1 / float(self.plan_speed_refer[1].split("hz")[0])
            speed = torch.linalg.norm(sp_traj, dim=-1).sum(-1) / (sp_mask.sum(-1) * interval + 1e-4)          # (B,)
            sp_any = sp_mask.any(dim=-1)
        for name, idx in intervals.items():
            areas = [types[i][2] for i in idx]
            bucket = torch.ones_like(speed, dtype=torch.long)
            for k, (lo, hi) in enumerate(areas):
                bucket = torch.where((speed >= lo) & (speed < hi), torch.full_like(bucket, k), bucket)
            contiguous = idx == list(range(idx[0], idx[0] + len(idx)))
            sel = (lambda t, idx=idx: t[:, idx[0]:idx[0] + len(idx)]) if contiguous else \
                (lambda t, idx=idx: t.index_select(1, _const(idx, logit).long()))
            s_logits.append(sel(logit))                                                                          # (B, K)
            s_best.append(torch.gather(sel(best), 1, bucket[:, None, None, None].expand(B, 1, ts, 2)).squeeze(1))  # (B, ts, 2)
            g_traj, g_mask = self._plan_gt(data, types[idx[0]])
            s_gt.append(g_traj); s_mask.append(g_mask); s_bucket.append(bucket)
        n_s = len(intervals)
        if n_s and len({t.shape[1] for t in s_logits}) != 1:
            return self._loss_plan_general(outs, data)        # intervals with different numbers of buckets
        if n_s:
            s_cls = self.loss_plan_cls(torch.cat(s_logits, 0), torch.cat(s_bucket, 0), weight=sp_any.repeat(n_s),
                                       layers=n_s * layers).reshape(n_s, layers)
        # one L1 over the trajectories of the aligned groups and of the speed intervals
        pred = [group_major(take(best))] + s_best
        tgt = [group_major(a_gt)] + s_gt
        msk = [group_major(a_mask)] + s_mask
        n_all = n_a + n_s
        all_reg = self.loss_plan_reg(torch.cat(pred, 0).cumsum(dim=-2), torch.cat(tgt, 0).cumsum(dim=-2),
                                     weight=torch.cat(msk, 0).unsqueeze(-1), layers=n_all * layers).reshape(n_all, layers)
        total = {}
        for j, i in enumerate(align):
            c, rg = total.setdefault(types[i][0], [0.0, 0.0])
            total[types[i][0]] = [c + a_cls[j], rg + all_reg[j]]
        for j in range(n_s):
            c, rg = total.setdefault("speed", [0.0, 0.0])
            total["speed"] = [c + s_cls[j], rg + all_reg[n_a + j]]
        kinds = []
        for t in types:
            if t[0] not in kinds:
                kinds.append(t[0])
        output = {}
        for k in kinds:
            self._add(output, f"plan_loss_{k}_cls", total[k][0])
            self._add(output, f"plan_loss_{k}_reg", total[k][1])
        return output

    def _loss_plan_general(self, outs, data):
        layers = len(outs["classification"])
        cls, reg = _stack_layers(outs["classification"]), _stack_layers(outs["prediction"])
        data = {k: (_tile(v, layers) if k.startswith("gt_ego_") else v) for k, v in data.items()}
        kinds = []
        for t in self.plan_anchor_types:
            if t[0] not in kinds:
                kinds.append(t[0])
        total = {k: [0.0, 0.0] for k in ("temp", "spat", "speed")}
        ref_cls, ref_reg = self._plan_slice(cls, reg, self.plan_anchor_refer)
        ref_gt, ref_mask = self._plan_gt(data, self.plan_anchor_refer)
        # the winning mode of the reference group decides the target mode of every other group
        _, ref_target, ref_weight, _, _, _ = self.plan_sampler.sample(ref_cls, ref_reg, ref_gt, ref_mask, data)
        speed = {}
        for kind in self.plan_anchor_types:
            p_cls, p_reg = self._plan_slice(cls, reg, kind)
            g_traj, g_mask = self._plan_gt(data, kind)
            if kind[0] in ("temp", "spat"):
                a_cls, _, _, best, g_t, g_m = self.align_sampler.sample(p_cls, p_reg, g_traj, g_mask, data, ref_target)
                c, r = self._plan_terms(a_cls, ref_target, ref_weight, best, g_t, g_m, layers)
                total[kind[0]][0] = total[kind[0]][0] + c
                total[kind[0]][1] = total[kind[0]][1] + r
            elif kind[0] == "speed":
                grp = speed.setdefault(kind[1], dict(cls=[], reg=[], gt=(g_traj, g_mask), areas=[]))
                grp["cls"].append(p_cls); grp["reg"].append(p_reg); grp["areas"].append(kind[2])
            else:
                raise NotImplementedError(kind)
        for grp in speed.values():
            c, r = self._speed_terms(grp, ref_target, data, layers)
            total["speed"][0] = total["speed"][0] + c
            total["speed"][1] = total["speed"][1] + r
        output = {}
        for k in kinds:
            self._add(output, f"plan_loss_{k}_cls", total[k][0])
            self._add(output, f"plan_loss_{k}_reg", total[k][1])
        return output

    def _speed_terms(self, grp, ref_target, data, layers):
        """Speed groups (reference :1371-1443): every speed bucket contributes the logit and the trajectory of the
        reference mode; the bucket of the ground-truth average speed is the class target."""
        g_traj, g_mask = grp["gt"]
        rows = torch.arange(ref_target.shape[0], device=ref_target.device)
        logits, trajs = [], []
        for p_cls, p_reg in zip(grp["cls"], grp["reg"]):
            a_cls, _, _, a_reg, _, _ = self.align_sampler.sample(p_cls, p_reg, g_traj, g_mask, data, ref_target)
            logits.append(a_cls.squeeze(1)[rows, ref_target.squeeze(-1)][:, None, None])
            trajs.append(a_reg[:, :, None])
        logits, trajs = torch.cat(logits, dim=-1), torch.cat(trajs, dim=-3)       # (bs,1,K), (bs,1,K,ts,2)
        sp_traj, sp_mask = self._plan_gt(data, self.plan_speed_refer)
        sp_traj, sp_mask = sp_traj.unsqueeze(1), sp_mask.unsqueeze(1)
        interval = 1 / float(self.plan_speed_refer[1].split("hz")[0])
        speed = torch.linalg.norm(sp_traj, dim=-1).sum(-1) / (sp_mask.sum(-1) * interval + 1e-4)
        bucket = torch.ones_like(speed, dtype=torch.long)
        for i, (lo, hi) in enumerate(grp["areas"]):
            bucket = torch.where((speed >= lo) & (speed < hi), torch.full_like(bucket, i), bucket)
        bs, _, k = logits.shape
        cls_k = logits.reshape(bs, self.ego_fut_cmd, 1, k)
        reg_k = trajs.reshape(bs, self.ego_fut_cmd, 1, k, self.ego_fut_ts, 2)
        if self.ego_fut_cmd > 1:
            cmd = data["gt_ego_fut_cmd"].argmax(dim=-1)
            cls_k, reg_k = cls_k[rows, cmd], reg_k[rows, cmd]
        else:
            cls_k, reg_k = cls_k[:, 0], reg_k[:, 0]
        return self._plan_terms(cls_k, bucket, sp_mask.any(dim=-1), _take_mode(reg_k, bucket),
                                g_traj.unsqueeze(1), g_mask.unsqueeze(1), layers)
