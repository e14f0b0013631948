"""Optimiser step of the training frame on flat buffers: global-norm clipping + AdamW in two kernel
launches (hip-ad_amd/csrc/optim.hip) instead of torch's ~170 multi-tensor launches over ~2000 tensors.

Reference: AdamW lr 2e-4, weight_decay 1e-3, ``paramwise_cfg`` lr_mult 0.5 for ``img_backbone``,
``grad_clip`` max_norm 25 (projects/configs/hipad_b2d_stage2.py:629-641).
"""
import math

import torch

from . import lib as _lib
from .dist import FlatGrads, flat_offsets


def lr_factor(lr_config, iteration, max_iters):
    """Closed form of the reference's learning-rate schedule at 0-based ``iteration`` (factor on the base lr).

    ``lr_config`` as in projects/configs/hipad_b2d_stage2.py:643-649: policy "CosineAnnealing", warmup "linear",
    warmup_iters 500, warmup_ratio 1/3, min_lr_ratio 1e-3, driven per iteration (IterBasedRunner).  Restates mmcv
    1.7.1 ``CosineAnnealingLrUpdaterHook.get_lr`` + ``LrUpdaterHook.get_warmup_lr`` (mmcv is not installed here, so
    this restatement is not pinned against mmcv itself); the device evaluates the same expression inside
    adamw_flat_kernel (optim.hip ``lr_factor``)."""
    if not lr_config:
        return 1.0
    policy = lr_config.get("policy")
    f = 1.0
    if policy == "CosineAnnealing":
        target = lr_config.get("min_lr_ratio", 0.0)
        x = min(iteration / max_iters, 1.0) if max_iters > 0 else 0.0
        f = target + 0.5 * (1.0 - target) * (1.0 + math.cos(math.pi * x))
    elif policy not in (None, "fixed", "Fixed"):
        raise NotImplementedError(f"lr policy {policy!r} (the HiP-AD configs use CosineAnnealing)")
    warm = lr_config.get("warmup")
    wi = lr_config.get("warmup_iters", 0)
    if warm is not None and iteration < wi:
        if warm != "linear":
            raise NotImplementedError(f"warmup {warm!r} (the HiP-AD configs use linear)")
        k = (1.0 - iteration / wi) * (1.0 - lr_config.get("warmup_ratio", 0.1))
        f *= 1.0 - k
    return f


def schedule_struct(lr_config, max_iters):
    """``lr_config`` dict -> the C struct hipad_adamw_step takes (None for a constant rate)."""
    if not lr_config:
        return None
    lr_factor(lr_config, 0, max_iters)  # validates policy / warmup names
    s = _lib.LrScheduleStruct()
    s.policy = 1 if lr_config.get("policy") == "CosineAnnealing" else 0
    s.warmup_iters = int(lr_config.get("warmup_iters", 0)) if lr_config.get("warmup") else 0
    s.warmup_ratio = float(lr_config.get("warmup_ratio", 0.1))
    s.max_iters = int(max_iters)
    s.min_lr_ratio = float(lr_config.get("min_lr_ratio", 0.0))
    return s


class FlatAdamW:
    """``groups`` = [(params, lr), (params, lr)] (one or two groups).  Every parameter becomes a view into
    ``self.flat_p`` (values preserved) and every gradient a view into ``self.grads.flat`` at the same offset;
    the moments are flat buffers too.  ``step()`` clips by the global norm and applies AdamW; the pre-clip
    norm is left in ``self.grad_norm`` (device scalar).  State lives on the device: capturable."""

    def __init__(self, groups, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=None, comm_dtype=None,
                 lr_config=None, max_iters=0, bf16_shadow=False, chain_operands=True):
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
        self._stats = torch.zeros(2, dtype=torch.float32, device=ref.device)  # [pre-clip norm, lr of group 0]
        self.grad_norm = self._stats[0:1]
        self.last_lr = self._stats[1:2]
        self._ws = torch.empty(_lib.load().hipad_adamw_workspace(), dtype=torch.uint8, device=ref.device)
        # learning-rate schedule: evaluated inside the kernel from the device step counter (capturable: a replayed
        # graph follows warm-up + cosine annealing without recapture or host updates)
        self.lr_config, self.max_iters = lr_config, max_iters
        self._sched = schedule_struct(lr_config, max_iters)
        # bf16 copy of the flat parameter buffer, kept current by the AdamW kernel (operands of the MFMA kernels)
        self.shadow = None
        self.offsets = offsets
        if bf16_shadow:
            self.shadow = torch.empty(total, dtype=torch.bfloat16, device=ref.device)
        self._pack = None
        if chain_operands:
            self._attach_chain_operands(ref.device)
        self.refresh_shadow()

    # ---- bf16 operand copies of the MLP-chain kernels -----------------------------------------------------
    # Every 2-D parameter small enough for the chain kernels (both dims <= 256) gets two bf16 copies in their operand
    # layout -- MFMA-fragment order of W and of W^T (include/hipad.h, hipad_pack_weights) -- refreshed from the fp32
    # master by ONE pack launch per step.  Each such parameter carries ``_hipad_shadow = (w, wt)``,
    # which hipad_amd.chain.bf16_pair prefers over its own version-keyed cache (the optimiser kernel updates
    # parameters through raw pointers, invisible to torch's version counters).
    def _attach_chain_operands(self, device):
        # (parameter, first row, rows): whole matrices with both dims <= 256, and the 256-row blocks of tall packed
        # projections (attention in_proj_weight [3 E][E], E <= 256) -- a row block of a row-major matrix is itself a
        # contiguous matrix, so it packs like one; those parameters carry ``_hipad_shadow_rows = {(r0, r1): (w, wt)}``
        mats = []
        for p in self.params:
            if p.dim() != 2 or p.shape[1] > 256:
                continue
            n = p.shape[0]
            if n <= 256:
                mats.append((p, 0, n))
            elif n % 256 == 0 and p.shape[1] == 256:
                mats += [(p, r0, 256) for r0 in range(0, n, 256)]
        if not mats:
            return
        from .chain import packed_numel
        total = sum(packed_numel(n, p.shape[1]) + packed_numel(p.shape[1], n) for p, _, n in mats)
        self.chain_buf = torch.zeros(total, dtype=torch.bfloat16, device=device)
        src, dst, dst_t, rows, cols, starts, tiles, off = [], [], [], [], [], [0], 0, 0
        for p, r0, n in mats:
            k = p.shape[1]
            w = self.chain_buf[off:off + packed_numel(n, k)]
            off += packed_numel(n, k)
            wt = self.chain_buf[off:off + packed_numel(k, n)]
            off += packed_numel(k, n)
            if n == p.shape[0]:
                p._hipad_shadow = (w, wt)
            else:
                if r0 == 0:
                    p._hipad_shadow_rows = {}
                p._hipad_shadow_rows[(r0, r0 + n)] = (w, wt)
            src.append(p.data_ptr() + 4 * r0 * k); dst.append(w.data_ptr()); dst_t.append(wt.data_ptr()); rows.append(n); cols.append(k)
            tiles += ((n + 31) // 32) * ((k + 31) // 32)
            starts.append(tiles)
        i64 = lambda v: torch.tensor(v, dtype=torch.int64, device=device)  # noqa: E731
        i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=device)  # noqa: E731
        self._pack = dict(src=i64(src), dst=i64(dst), dst_t=i64(dst_t), rows=i32(rows), cols=i32(cols), starts=i32(starts),
                          n=len(mats), tiles=tiles)

    def _pack_chain_operands(self):
        pk = self._pack
        if pk is None:
            return
        lib = _lib.load()
        dev = self.flat_p.device
        with torch.cuda.device(dev):
            _lib.check(lib.hipad_pack_weights(pk["dst"].data_ptr(), pk["dst_t"].data_ptr(), pk["src"].data_ptr(),
                                              pk["rows"].data_ptr(), pk["cols"].data_ptr(), pk["starts"].data_ptr(), pk["n"],
                                              pk["tiles"], _lib.stream_ptr(dev)), "hipad_pack_weights")

    def refresh_shadow(self):
        """Re-derive the bf16 copies from the fp32 parameters.  Runs by itself when a consumer of a copy finds that torch
        wrote the parameter since the copies were made (load_state_dict, checkpoint resume, ``p.copy_(...)``): every
        parameter is stamped with its version counter here and after every ``step()`` (the optimiser kernel itself
        updates parameters through raw pointers and does not move the counters), and ``shadow_is_current`` compares."""
        if self.shadow is not None:
            _lib.shadow_bf16(self.shadow, self.flat_p)
        self._pack_chain_operands()
        self._stamp()

    def _stamp(self):
        import weakref
        ref = weakref.WeakMethod(self.refresh_shadow)
        for p in self.params:
            p._hipad_shadow_stamp = (p._version, p.data_ptr(), ref)

    def shadow_of(self, p):
        """bf16 view of parameter ``p`` inside the shadow buffer (None when no shadow is kept)."""
        if self.shadow is None:
            return None
        off = (p.data_ptr() - self.flat_p.data_ptr()) // 4
        if not (0 <= off < self.flat_p.numel()):
            return None
        return self.shadow[off:off + p.numel()].view(p.shape)

    def lr_at(self, iteration):
        """(lr group 0, lr group 1) the kernel uses at 0-based ``iteration`` (host closed form)."""
        f = lr_factor(self.lr_config, iteration, self.max_iters)
        return self.lrs[0] * f, self.lrs[1] * f

    def step(self, zero_grad=True):
        _lib.adamw_step(self.flat_p, self.grads.flat, self.exp_avg, self.exp_avg_sq, self.n_group0, self.lrs[0],
                        self.lrs[1], self.betas, self.eps, self.weight_decay, self.max_norm, self.step_count,
                        self._stats, self._ws, zero_grad=zero_grad, sched=self._sched, shadow=self.shadow)
        self._pack_chain_operands()


def shadow_is_current(weight):
    """True when the bf16 copies an optimiser attached to ``weight`` (``_hipad_shadow``, ``_hipad_shadow_rows``,
    ``_hipad_bf16``) may be used: torch has not written the parameter since they were made -- or it has, and the owning
    optimiser could be asked to re-derive them (done here).  False: the caller falls back to deriving its own copy.
    Replacing a parameter's storage (``p.data = other``) detaches it from the flat buffers: that is an error."""
    stamp = getattr(weight, "_hipad_shadow_stamp", None)
    if stamp is None:
        return True                      # copies attached by hand (tests): the caller's responsibility
    version, ptr, refresh = stamp
    if weight.data_ptr() != ptr:
        raise RuntimeError("a parameter managed by FlatAdamW was given new storage (p.data = ...): its flat-buffer views and "
                           "bf16 operand copies no longer follow it; write into it with p.copy_() / load_state_dict instead")
    if weight._version == version:
        return True
    fn = refresh()
    if fn is None:
        return False
    fn()                                 # re-derives and re-stamps every copy of that optimiser
    return True
