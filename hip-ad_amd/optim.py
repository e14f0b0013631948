"""Optimiser step of the training frame on flat buffers: global-norm clipping + AdamW in two kernel
launches (hip-ad_amd/csrc/optim.hip) instead of torch's ~170 multi-tensor launches over ~2000 tensors.

Reference: AdamW lr 2e-4, weight_decay 1e-3, ``paramwise_cfg`` lr_mult 0.5 for ``img_backbone``,
``grad_clip`` max_norm 25 (projects/configs/hipad_b2d_stage2.py:629-641).
"""
import torch

from . import lib as _lib
from .dist import FlatGrads, flat_offsets


class FlatAdamW:
    """``groups`` = [(params, lr), (params, lr)] (one or two groups).  Every parameter becomes a view into
    ``self.flat_p`` (values preserved) and every gradient a view into ``self.grads.flat`` at the same offset;
    the moments are flat buffers too.  ``step()`` clips by the global norm and applies AdamW; the pre-clip
    norm is left in ``self.grad_norm`` (device scalar).  State lives on the device: capturable."""

    def __init__(self, groups, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=None, comm_dtype=None):
        if not 1 <= len(groups) <= 2:
            raise ValueError("one or two learning-rate groups")
        plists = [[p for p in ps if p.requires_grad] for ps, _ in groups]
        self.params = [p for ps in plists for p in ps]
        ref = self.params[0]
        if any(p.dtype != torch.float32 or not p.is_contiguous() for p in self.params):
            raise ValueError("FlatAdamW needs contiguous fp32 parameters")
        self.grads = FlatGrads(self.params, comm_dtype=comm_dtype)
        offsets, total = flat_offsets(self.params)
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=ref.device)
        with torch.no_grad():
            for p, off in zip(self.params, offsets):
                view = self.flat_p[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
        self.n_group0 = offsets[len(plists[0])] if len(plists) == 2 and plists[1] else total
        self.lrs = (groups[0][1], groups[1][1] if len(groups) == 2 else groups[0][1])
        self.betas, self.eps, self.weight_decay, self.max_norm = betas, eps, weight_decay, max_norm
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.step_count = torch.zeros(1, dtype=torch.int32, device=ref.device)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=ref.device)
        self._ws = torch.empty(_lib.load().hipad_adamw_workspace(), dtype=torch.uint8, device=ref.device)

    def step(self, zero_grad=True):
        _lib.adamw_step(self.flat_p, self.grads.flat, self.exp_avg, self.exp_avg_sq, self.n_group0, self.lrs[0],
                        self.lrs[1], self.betas, self.eps, self.weight_decay, self.max_norm, self.step_count,
                        self.grad_norm, self._ws, zero_grad=zero_grad)
