"""CPU restatement of the whole training frame -- TEST INFRASTRUCTURE ONLY (bench.py's ``cpu_baseline`` leg and tests).

The product has no CPU path behind its kernels (a CPU tensor raises).  For the CPU baseline of BASELINE.md section 3 --
"the reference's CPU-only PyTorch path timed on the same box's host cores" -- this module runs the mirrored modules of
``projects/mmdet3d_plugin`` on CPU tensors (their Linear / LayerNorm / convolution layers are plain torch there) and
supplies, for the duration of a ``with cpu_path():`` block, CPU restatements of the four operators that exist only as HIP
kernels:

  deformable aggregation  -> oracle/daf_oracle.c (the C restatement of the reference CUDA kernels, anchors dealt to
                             ``threads`` host threads; reference ops/src/deformable_aggregation_cuda.cu:129-262)
  attention core          -> softmax(q k^T / sqrt(d)) v in fp32 torch (reference models/attention.py:36-98 calls flash-attn)
  sampling weights        -> reference models/blocks.py:178-214 (_get_weights: joint softmax over cams x levels x points)
  3D -> 2D projection     -> reference models/blocks.py:216-225 (project_points)

Nothing here is imported by the product path.
"""
import contextlib
import math
import os
import time

import numpy as np
import torch
from torch.autograd.function import Function, once_differentiable

from . import daf as O


class _OracleDAF(Function):
    @staticmethod
    def forward(ctx, feat, spatial_shape, scale_start_index, loc, weights):
        f, l, w = (t.detach().contiguous().float().numpy() for t in (feat, loc, weights))
        ss, st = spatial_shape.int().numpy(), scale_start_index.int().numpy()
        ctx.saved = (f, ss, st, l, w)
        return torch.from_numpy(O.daf_forward(f, ss, st, l, w))

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        f, ss, st, l, w = ctx.saved
        gf, gl, gw = O.daf_backward(f, ss, st, l, w, gout.contiguous().float().numpy())
        return torch.from_numpy(gf), None, None, torch.from_numpy(gl), torch.from_numpy(gw)


def daf(feat, spatial_shape, scale_start_index, loc, weights):
    return _OracleDAF.apply(feat, spatial_shape, scale_start_index, loc, weights)


def attention(q, k, v, heads, scale=None, p_drop=0.0, seed=0):
    B, Nq, E = q.shape
    D = E // heads
    scale = D ** -0.5 if scale is None else scale

    def split(t):
        return t.reshape(B, t.shape[1], heads, D).transpose(1, 2)

    att = torch.softmax((split(q) @ split(k).transpose(-1, -2)) * scale, dim=-1)
    if p_drop > 0.0:
        att = torch.nn.functional.dropout(att, p_drop)
    return (att @ split(v)).transpose(1, 2).reshape(B, Nq, E)


def sampling_weights(u, v, keep, L, P, G):
    """(bs, A, P, cams, L, G): softmax over (cams, L, P) per group of u[b, a] + v[b, cam] (layout ((l P + p) G + g))."""
    if v is not None:
        logits = u[:, :, None] + v[:, None]
    else:
        logits = u
    bs, A, cams = logits.shape[:3]
    w = logits.reshape(bs, A, cams * L * P, G).softmax(dim=-2).reshape(bs, A, cams, L, P, G)
    if keep is not None:
        w = w * keep[:, :, :, None, :, None]
    return w.permute(0, 1, 4, 2, 3, 5).contiguous()


def project_points(key_points, projection_mat, image_wh=None):
    pts = torch.cat([key_points, torch.ones_like(key_points[..., :1])], dim=-1)
    p = torch.matmul(projection_mat[:, :, None, None], pts[:, None, ..., None]).squeeze(-1)   # (bs, cams, A, P, 4)
    uv = p[..., :2] / torch.clamp(p[..., 2:3], min=1e-5)
    if image_wh is not None:
        uv = uv / image_wh[:, :, None, None]
    return uv.permute(0, 2, 3, 1, 4).contiguous()


@contextlib.contextmanager
def cpu_path(threads=None):
    """Patch the four HIP-only operators with the CPU restatements above (and restore them afterwards)."""
    from hipad_amd import functional as HF
    import importlib
    blocks = importlib.import_module("projects.mmdet3d_plugin.models.blocks")
    # a one-GPU box owns a 16-core share of its host (256 logical CPUs are visible); HIPAD_CPU_THREADS overrides
    threads = threads or int(os.environ.get("HIPAD_CPU_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    saved = (HF.attention, HF.sampling_weights, HF.project_points, blocks.DAF, torch.get_num_threads(), O.lib().hipad_oracle_get_threads())
    HF.attention, HF.sampling_weights, HF.project_points, blocks.DAF = attention, sampling_weights, project_points, daf
    torch.set_num_threads(threads)
    O.set_threads(threads)
    try:
        yield threads
    finally:
        HF.attention, HF.sampling_weights, HF.project_points, blocks.DAF = saved[:4]
        torch.set_num_threads(saved[4])
        O.set_threads(saved[5])


def time_frames(seconds=20.0, plan_queries=480, threads=None, input_hw=(256, 704), seed=0):
    """Whole stage-2 training frames (encoder + decoder + losses, forward + backward, fp32) on the host cores until
    ``seconds`` have passed (at least one timed frame after one untimed temporal warm-up frame)."""
    import warnings
    warnings.filterwarnings("ignore")
    from hipad_amd.frame import SyntheticFrames, build_detector, frame_losses
    with cpu_path(threads) as used:
        torch.manual_seed(1234)
        model, _ = build_detector(stage=2, input_hw=input_hw, plan_queries=plan_queries, device="cpu")
        model.encoder_dtype = torch.float32
        model.train()
        frames = SyntheticFrames(bs=1, input_hw=input_hw, device="cpu", seed=seed)

        def one():
            img, data = frames.next()
            losses = frame_losses(model, img, data)
            total = sum(losses.values())
            model.zero_grad(set_to_none=True)
            total.backward()
            return float(total)

        one()   # cold frame (no temporal cache yet): untimed
        spent, n, last = 0.0, 0, None
        while n < 1 or spent < seconds:
            t = time.perf_counter()
            last = one()
            spent += time.perf_counter() - t
            n += 1
    return dict(frames=n, seconds=spent, cores=used, loss=last)
