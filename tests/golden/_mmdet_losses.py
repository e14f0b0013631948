"""Stand-ins for the mmdet==2.28.2 loss / matching primitives the reference's loss path imports (BUILD CONTAINER
ONLY, used by make_golden.py through _ref_shim.py).  mmdet is a third-party package that is neither vendored in the
reference nor installed here; these restate its published formulas in mmdet's own calling conventions
(``weighted_loss`` / ``weight_reduce_loss`` semantics, reduction='mean' with the eps-guarded avg_factor of 2.28):

  FocalLoss(use_sigmoid)   mmdet/models/losses/focal_loss.py        py_sigmoid_focal_loss
  L1Loss, l1_loss, smooth_l1_loss  mmdet/models/losses/smooth_l1_loss.py
  CrossEntropyLoss(use_sigmoid)    mmdet/models/losses/cross_entropy_loss.py  binary_cross_entropy
  GaussianFocalLoss        mmdet/models/losses/gaussian_focal_loss.py
  FocalLossCost            mmdet/core/bbox/match_costs/match_cost.py
Parity of these five against mmdet itself is UNPINNED (no mmdet, no reference fixture): what the golden vectors pin
is the reference's own target assignment / loss composition on top of them.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def weight_reduce_loss(loss, weight=None, reduction="mean", avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return {"none": loss, "mean": loss.mean(), "sum": loss.sum()}[reduction] if reduction != "none" else loss
    if reduction == "mean":
        eps = torch.finfo(torch.float32).eps
        return loss.sum() / (avg_factor + eps)
    if reduction == "none":
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


def l1_loss(pred, target, weight=None, reduction="mean", avg_factor=None):
    if target.numel() == 0:
        return pred.sum() * 0
    return weight_reduce_loss(torch.abs(pred - target), weight, reduction, avg_factor)


def smooth_l1_loss(pred, target, weight=None, beta=1.0, reduction="mean", avg_factor=None):
    if target.numel() == 0:
        return pred.sum() * 0
    diff = torch.abs(pred - target)
    loss = torch.where(diff < beta, 0.5 * diff * diff / beta, diff - 0.5 * beta)
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


class L1Loss(nn.Module):
    def __init__(self, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        if weight is not None and not torch.any(weight > 0):
            if pred.dim() == weight.dim() + 1:
                weight = weight.unsqueeze(1)
            return (pred * weight).sum()
        return self.loss_weight * l1_loss(pred, target, weight, reduction=reduction_override or self.reduction,
                                          avg_factor=avg_factor)


class FocalLoss(nn.Module):
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction="mean", loss_weight=1.0, activated=False):
        super().__init__()
        assert use_sigmoid and not activated
        self.gamma, self.alpha, self.reduction, self.loss_weight = gamma, alpha, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        num_classes = pred.size(1)
        target = F.one_hot(target, num_classes=num_classes + 1)[:, :num_classes]
        pred_sigmoid = pred.sigmoid()
        target = target.type_as(pred)
        pt = (1 - pred_sigmoid) * target + pred_sigmoid * (1 - target)
        focal_weight = (self.alpha * target + (1 - self.alpha) * (1 - target)) * pt.pow(self.gamma)
        loss = F.binary_cross_entropy_with_logits(pred, target, reduction="none") * focal_weight
        if weight is not None:
            if weight.shape != loss.shape:
                if weight.size(0) == loss.size(0):
                    weight = weight.view(-1, 1)
                else:
                    weight = weight.view(loss.size(0), -1)
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction_override or self.reduction, avg_factor)


class CrossEntropyLoss(nn.Module):
    def __init__(self, use_sigmoid=False, use_mask=False, reduction="mean", class_weight=None, ignore_index=None,
                 loss_weight=1.0, avg_non_ignore=False):
        super().__init__()
        assert use_sigmoid and not use_mask and class_weight is None
        self.reduction, self.loss_weight, self.ignore_index = reduction, loss_weight, ignore_index

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        ignore_index = -100 if self.ignore_index is None else self.ignore_index
        assert cls_score.dim() == label.dim()
        valid_mask = ((label >= 0) & (label != ignore_index)).float()
        weight = valid_mask if weight is None else weight * valid_mask
        loss = F.binary_cross_entropy_with_logits(cls_score, label.float(), pos_weight=None, reduction="none")
        return self.loss_weight * weight_reduce_loss(loss, weight.float(), reduction_override or self.reduction, avg_factor)


class GaussianFocalLoss(nn.Module):
    def __init__(self, alpha=2.0, gamma=4.0, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.alpha, self.gamma, self.reduction, self.loss_weight = alpha, gamma, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        eps = 1e-12
        pos_weights = target.eq(1)
        neg_weights = (1 - target).pow(self.gamma)
        pos_loss = -(pred + eps).log() * (1 - pred).pow(self.alpha) * pos_weights
        neg_loss = -(1 - pred + eps).log() * pred.pow(self.alpha) * neg_weights
        return self.loss_weight * weight_reduce_loss(pos_loss + neg_loss, weight, reduction_override or self.reduction,
                                                     avg_factor)


class FocalLossCost:
    def __init__(self, weight=1.0, alpha=0.25, gamma=2, eps=1e-12, binary_input=False):
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps

    def __call__(self, cls_pred, gt_labels):
        cls_pred = cls_pred.sigmoid()
        neg_cost = -(1 - cls_pred + self.eps).log() * (1 - self.alpha) * cls_pred.pow(self.gamma)
        pos_cost = -(cls_pred + self.eps).log() * self.alpha * (1 - cls_pred).pow(self.gamma)
        return (pos_cost[:, gt_labels] - neg_cost[:, gt_labels]) * self.weight


class BaseAssigner:
    pass


class AssignResult:
    pass
