"""Frame roofline of the training step: sum over the step's operators of max(bytes / HBM bandwidth, flops / MFMA peak)
("end-to-end frame roofline"), from the EXACT module shapes of one eager forward.

A census pass runs one forward of the model with hooks on every Conv2d / Linear / LayerNorm / BatchNorm module and
wrappers around the attention and aggregation operators, records (flops, algorithmic bytes) per call, and scales them
to forward + backward (x3 for GEMM-shaped work: dX and dW; the backbone's first convolution has no dX).  Peaks from
/opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s, dense bf16 MFMA 2.5 PFLOP/s.  Host-side measurement plumbing
only (bench.py); nothing here is on the product path.
"""
import collections

import torch
import torch.nn as nn

HBM_BPS = 8.0e12
MFMA_BF16_FLOPS = 2.5e15


def census(model, img, data, daf_bytes):
    """One eager forward (+ losses) of ``model`` with per-operator accounting.

    ``daf_bytes(A, P, kind)`` -> algorithmic bytes of an aggregation call of that shape (kind in fwd / bwd_lw / bwd_feat).
    Returns {category: dict(calls, flops, bytes)} for forward + backward of ONE frame."""
    import importlib
    from . import functional as HF
    from .frame import frame_losses
    blocks = importlib.import_module("projects.mmdet3d_plugin.models.blocks")
    acc = collections.defaultdict(lambda: dict(calls=0, flops=0.0, bytes=0.0))

    def add(cat, flops, nbytes):
        a = acc[cat]
        a["calls"] += 1
        a["flops"] += flops
        a["bytes"] += nbytes

    first_conv = [True]
    handles = []

    def conv_hook(mod, inp, out):
        x = inp[0]
        kh, kw = mod.kernel_size
        flops = 2.0 * out.numel() * (mod.in_channels // mod.groups) * kh * kw
        es = x.element_size()
        nbytes = (x.numel() + out.numel()) * es + mod.weight.numel() * es
        mult = 2.0 if first_conv[0] else 3.0      # forward + dW (+ dX unless it is the image itself)
        first_conv[0] = False
        cat = "encoder conv (MFMA)" if x.shape[-1] > 1 and x.dim() == 4 and x.shape[0] >= 6 else "decoder conv (MFMA)"
        add(cat, mult * flops, mult * nbytes)

    def norm_hook(mod, inp, out):
        x = inp[0]
        cat = "batch norm (HBM)" if isinstance(mod, nn.BatchNorm2d) else "layer norm (HBM)"
        add(cat, 0.0, 5.0 * x.numel() * x.element_size())   # fwd read+write, bwd read x, dy, write dx

    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            handles.append(m.register_forward_hook(conv_hook))
        elif isinstance(m, (nn.LayerNorm, nn.BatchNorm2d)):
            handles.append(m.register_forward_hook(norm_hook))

    real_attn, real_daf, real_linear = HF.attention, blocks.DAF, HF.linear

    def attn(q, k, v, heads, scale=None, p_drop=0.0, seed=0):
        B, Nq, E = q.shape
        Nk = k.shape[1]
        flops = 4.0 * B * Nq * Nk * E
        add("attention core (MFMA)", 3.5 * flops, 4.0 * 3.0 * (2 * B * Nq * E + 2 * B * Nk * E))
        return real_attn(q, k, v, heads, scale, p_drop, seed)

    def daf(*args):
        loc = args[-2]
        A, P = loc.shape[1:3]
        for kind in ("fwd", "bwd_lw", "bwd_feat"):
            add("aggregation gather / scatter (HBM)", 0.0, float(daf_bytes(loc, kind)))
        w = args[-1]
        add("sampling weights softmax + projection (HBM)", 0.0, 4.0 * w.numel() * 4 + 4.0 * loc.numel() * 4)
        return real_daf(*args)

    def flin(x, weight, bias=None, relu=False, rows=None):   # every Linear of the decoder goes through HF.linear
        r0, r1 = (0, weight.shape[0]) if rows is None else rows
        M = x.numel() // x.shape[-1]
        K = weight.shape[1]
        add("linear (MFMA)", 3.0 * 2.0 * M * (r1 - r0) * K, 3.0 * (4.0 * (M * K + M * (r1 - r0)) + 2.0 * (r1 - r0) * K))
        return real_linear(x, weight, bias, relu, rows)

    was_chains = HF.USE_CHAINS
    HF.attention, blocks.DAF, HF.linear, HF.USE_CHAINS = attn, daf, flin, False
    try:
        with torch.no_grad():
            dec = model.head.onedecoder_head
            step = dec.run_step
            frame_losses(model, img, data)
            dec.run_step = step
    finally:
        HF.attention, blocks.DAF, HF.linear, HF.USE_CHAINS = real_attn, real_daf, real_linear, was_chains
        for h in handles:
            h.remove()
    n_params = sum(p.numel() for p in model.parameters() if p.requires_grad)
    add("clip + AdamW (HBM)", 0.0, 36.0 * n_params)          # g read twice, p / m / v read + written
    return dict(acc)


def frame_roofline(categories, measured_ms):
    """Sum over categories of max(bytes / BW, flops / peak) -> dict for bench.py's JSON line."""
    rows, total = {}, 0.0
    for cat, a in sorted(categories.items()):
        t_mem, t_mma = a["bytes"] / HBM_BPS, a["flops"] / MFMA_BF16_FLOPS
        t = max(t_mem, t_mma)
        total += t
        rows[cat] = dict(calls=a["calls"], gflop=round(a["flops"] / 1e9, 2), mbytes=round(a["bytes"] / 1e6, 1),
                         roofline_ms=round(1e3 * t, 4), bound="mfma" if t_mma > t_mem else "hbm")
    return dict(roofline_ms=round(1e3 * total, 3), measured_ms=round(measured_ms, 3),
                frac=round(1e3 * total / measured_ms, 4) if measured_ms > 0 else None,
                peaks=dict(hbm_GBs=HBM_BPS / 1e9, mfma_bf16_TFLOPs=MFMA_BF16_FLOPS / 1e12), categories=rows,
                note="sum over operators of max(algorithmic bytes / 8 TB/s, flops / 2.5 PFLOP/s) for forward + backward + "
                     "optimiser of one frame; elementwise glue, loss arithmetic and launch gaps count as zero work")
