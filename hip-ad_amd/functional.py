"""autograd bindings of the HIP kernels (host-side plumbing: torch supplies memory + autograd).

  project_points(key_points, projection_mat, image_wh) -> loc (bs,A,P,cams,2)
  sampling_weights(u, v, keep, L, P, G)                -> weights (bs,A,P,cams,L,G)
  attention / layer_norm / linear wrappers live in their own sections below as they land.
"""
import torch
from torch.autograd.function import Function, once_differentiable

from . import lib as _lib


def _c32(t):
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


class _ProjectPoints(Function):
    @staticmethod
    def forward(ctx, key_points, projection_mat, image_wh):
        kp, pm, wh = _c32(key_points), _c32(projection_mat), _c32(image_wh)
        ctx.save_for_backward(kp, pm, wh)
        return _lib.project_points_forward(kp, pm, wh)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_loc):
        kp, pm, wh = ctx.saved_tensors
        gkp = _lib.project_points_backward(_c32(grad_loc), kp, pm, wh) if ctx.needs_input_grad[0] else None
        return gkp, None, None


def project_points(key_points, projection_mat, image_wh=None):
    """3D key points -> normalised image coordinates in the aggregation op's layout.

    key_points (bs,A,P,3), projection_mat (bs,cams,4,4), image_wh (bs,cams,2) or None
    -> (bs,A,P,cams,2).  Same arithmetic as the reference's project_points
    (projects/mmdet3d_plugin/models/blocks.py:216-225) + its permute (blocks.py:144-145).
    """
    return _ProjectPoints.apply(key_points, projection_mat, image_wh)


class _SamplingWeights(Function):
    @staticmethod
    def forward(ctx, u, v, keep, L, P, G):
        u, v, keep = _c32(u), _c32(v), _c32(keep)
        w, stats = _lib.weights_softmax_forward(u, v, keep, L, P, G)
        ctx.save_for_backward(u, v, keep, stats)
        ctx.dims = (L, P, G)
        return w

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_w):
        u, v, keep, stats = ctx.saved_tensors
        gu, gv = _lib.weights_softmax_backward(_c32(grad_w), stats, u, v, keep, *ctx.dims)
        return gu, gv, None, None, None, None


def sampling_weights(u, v, keep, L, P, G):
    """softmax over (cams, levels, points) of u[b,a,:] + v[b,cam,:] per group, written in the op
    layout (bs,A,P,cams,L,G).  u may also be (bs,A,cams,n) with v=None.  keep: (bs,A,cams,P) or None."""
    return _SamplingWeights.apply(u, v, keep, L, P, G)


_DROPOUT_CLOCK = {}
_CALL_SITE = [0]


def dropout_clock(device):
    """Device-resident int32 step counter mixed into the attention dropout seed.  ``advance`` it once
    per training step (an in-place add: capturable), so a replayed graph draws fresh masks."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    t = _DROPOUT_CLOCK.get(key)
    if t is None:
        t = _DROPOUT_CLOCK[key] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", key))
    return t


def advance_dropout_clock(device):
    dropout_clock(device).add_(1)


def new_call_site_seed():
    """A distinct constant per attention module, so modules draw different masks at the same step."""
    _CALL_SITE[0] += 1
    return (_CALL_SITE[0] * 2654435761) & 0x7FFFFFFF


class _Attention(Function):
    @staticmethod
    def forward(ctx, q, k, v, heads, scale, p_drop, seed):
        q, k, v = _c32(q), _c32(k), _c32(v)
        need = any(ctx.needs_input_grad[:3])
        clock = dropout_clock(q.device) if p_drop > 0.0 else None
        out, lse = _lib.attention_forward(q, k, v, heads, scale, p_drop, seed, need_lse=need, seed_dev=clock)
        if need:
            ctx.save_for_backward(q, k, v, out, lse)
        ctx.cfg = (heads, scale, p_drop, seed)
        ctx.clock = clock
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        dq, dk, dv = _lib.attention_backward(_c32(dout), out, lse, q, k, v, *ctx.cfg, seed_dev=ctx.clock)
        return dq, dk, dv, None, None, None, None


def attention(q, k, v, heads, scale=None, p_drop=0.0, seed=0):
    """softmax(q k^T * scale) v per head on (B, N, heads*D) tensors (see include/hipad.h).  With
    dropout the mask depends on (seed, dropout_clock): pass a per-module constant as ``seed`` and
    advance the clock once per step."""
    if scale is None:
        scale = (q.shape[-1] // heads) ** -0.5
    return _Attention.apply(q, k, v, heads, float(scale), float(p_drop), int(seed))


class _ZeroArena:
    """Zero-initialised fp32 scratch handed out in slices and cleared by ONE fill per frame: the partial-sum buffers of
    the BatchNorm kernels must be zero on entry, and a torch.zeros per layer would cost the launch the fusion saves.
    ``reset()`` (SparseDetector.extract_feat, before the encoder runs) clears everything handed out since the last reset;
    a slice stays untouched until the call that consumes it -- the backward's slice is reserved during the forward."""

    def __init__(self):
        self.buf, self.used, self.high = None, 0, 0
        self.generation = 0     # bumped by every reset(): a slice handed out before the reset may have been handed out again

    def reset(self, device, capacity=1 << 20):
        if self.buf is None or self.buf.device != torch.device(device) or self.buf.numel() < max(capacity, self.high):
            self.buf = torch.zeros(max(capacity, 2 * self.high), dtype=torch.float32, device=device)
        else:
            self.buf[:max(self.used, 1)].zero_()
        self.used = 0
        self.generation += 1

    def take(self, n, device):
        n = (n + 63) // 64 * 64
        if self.buf is None or self.buf.device != torch.device(device) or self.used + n > self.buf.numel():
            self.high = max(self.high, self.used + n)
            return torch.zeros(n, dtype=torch.float32, device=device)     # outside a reset() cycle / arena too small
        out = self.buf[self.used:self.used + n]
        self.used += n
        self.high = max(self.high, self.used)
        return out


BN_ARENA = _ZeroArena()


class _BatchNormAct(Function):
    """relu?(batch_norm(x) (+ residual)) on the two fused kernels (csrc/batchnorm.hip); x bf16 channels-last."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, eps, momentum, relu, out=None):
        c = x.shape[1]
        n = _lib.BN_REPLICAS * 2 * c * _lib.BN_SUM_FLOATS
        sums = BN_ARENA.take(n, x.device)
        ctx.gsums = BN_ARENA.take(n, x.device) if any(ctx.needs_input_grad[:4]) else None
        ctx.arena_generation = BN_ARENA.generation
        y, save = _lib.bn_forward(x, residual, weight.detach(), bias.detach(), running_mean, running_var, sums, eps, momentum, relu,
                                  out=out)
        ctx.grouped = out is not None
        if out is not None:
            # y was written into `out` (groups, rows, C) -- a level's block of the flat pyramid, one group per sample; hand
            # it on as the (samples, cameras, C, h, w) level it is
            if relu:
                raise _lib.HipadError("batch_norm_act: grouped output is for the FPN's last norm layer (no ReLU)")
            n, _, h, w = x.shape
            y = out.view(out.shape[0], n // out.shape[0], h, w, c).permute(0, 1, 4, 2, 3)
        ctx.save_for_backward(x, y if relu else None, save)
        ctx.weight, ctx.bias, ctx.has_res = weight, bias, residual is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, y, save = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        if ctx.grouped:
            dy = dy.flatten(0, 1)                   # (samples, cameras, C, h, w) -> (N, C, h, w)
        if dy.dtype != x.dtype or not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.to(x.dtype).contiguous(memory_format=torch.channels_last)
        rets = [None, None]
        targets = []
        for i, p in enumerate((weight, bias)):
            g = p.grad if (LINEAR_INPLACE_GRAD and p.is_leaf and ctx.needs_input_grad[1 + i]) else None
            if g is not None and g.is_contiguous() and g.dtype == torch.float32:
                INPLACE_PARAMS.add(id(p))
                targets.append(g)
            elif ctx.needs_input_grad[1 + i]:
                rets[i] = torch.zeros_like(p, dtype=torch.float32)
                targets.append(rets[i])
            else:
                targets.append(None)
        gsums, ctx.gsums = ctx.gsums, None          # the reserved slice is zero only once (retain_graph: fresh zeros after)
        if gsums is not None and ctx.arena_generation != BN_ARENA.generation:
            gsums = None    # another training forward reset the arena since ours (gradient accumulation, an auxiliary
            #                 forward): the reserved slice may hold that forward's sums by now -- take fresh zeros
        if gsums is None:
            gsums = torch.zeros(_lib.BN_REPLICAS * 2 * x.shape[1] * _lib.BN_SUM_FLOATS, dtype=torch.float32, device=x.device)
        dx, dres = _lib.bn_backward(dy, y, x, save, weight.detach(), gsums, targets[0], targets[1],
                                    ctx.has_res and ctx.needs_input_grad[3])
        return dx if ctx.needs_input_grad[0] else None, rets[0], rets[1], dres, None, None, None, None, None, None


def batch_norm_act_ok(x, weight):
    """Inputs the fused BatchNorm kernels take: bf16 channels-last (N, C, H, W) on the GPU, C a supported width."""
    return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and weight is not None and weight.dtype == torch.float32
            and x.is_contiguous(memory_format=torch.channels_last)
            and _lib.bn_supported(x.shape[0] * x.shape[2] * x.shape[3], x.shape[1]))


def batch_norm_act(x, weight, bias, running_mean, running_var, eps, momentum, relu=False, residual=None, out=None):
    if residual is not None and (residual.dtype != x.dtype or residual.shape != x.shape
                                 or not residual.is_contiguous(memory_format=torch.channels_last)):
        residual = residual.to(x.dtype).contiguous(memory_format=torch.channels_last)
    return _BatchNormAct.apply(x, weight, bias, residual, running_mean, running_var, float(eps), float(momentum), bool(relu), out)


class _FlatPyramid(Function):
    """The flat pyramid tensor whose blocks the FPN's last norm layers have ALREADY written (batch_norm_act(out=...)):
    nothing to compute forward -- the node only tells autograd that ``flat`` is made of the levels, and hands each level
    its block of the flat gradient in the backward (views, no copy)."""

    @staticmethod
    def forward(ctx, flat, blocks, *levels):
        ctx.blocks = blocks                               # [(first row, rows per sample)] per level
        ctx.shapes = [tuple(t.shape) for t in levels]
        for t, (off, n) in zip(levels, blocks):
            if t.data_ptr() != flat.data_ptr() + off * flat.shape[2] * flat.element_size():
                raise _lib.HipadError("flat pyramid: a level is not the block of the flat tensor it claims to be")
        return flat.view_as(flat)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        outs = []
        for (off, n), (bs, cams, c, h, w) in zip(ctx.blocks, ctx.shapes):
            outs.append(g[:, off:off + n].view(bs, cams, h, w, c).permute(0, 1, 4, 2, 3))
        return (None, None) + tuple(outs)


def flat_pyramid(flat, blocks, levels):
    return _FlatPyramid.apply(flat, blocks, *levels)


class _DropoutAdd(Function):
    @staticmethod
    def forward(ctx, x, identity, p_drop, seed):
        ctx.cfg = (p_drop, seed, dropout_clock(x.device))
        return _lib.dropout_add(_c32(x), _c32(identity), p_drop, seed, ctx.cfg[2])

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        p_drop, seed, clock = ctx.cfg
        dout = _c32(dout)
        dx = _lib.dropout_add(dout, None, p_drop, seed, clock) if ctx.needs_input_grad[0] else None
        return dx, (dout if ctx.needs_input_grad[1] else None), None, None


def dropout_add(x, identity, p_drop, seed, training=True):
    """identity + dropout(x) as one launch (backward: one launch for dx, the identity branch passes the gradient on);
    the mask is a function of (seed, dropout clock, element index) -- pass a per-module constant as ``seed``."""
    if not training or p_drop <= 0.0:
        return identity + x
    if not (x.is_cuda and x.dtype == torch.float32 and identity.dtype == torch.float32 and x.shape == identity.shape
            and x.numel() % 4 == 0):
        return identity + torch.nn.functional.dropout(x, p_drop, True)
    return _DropoutAdd.apply(x, identity, float(p_drop), int(seed))


import os as _os

LINEAR_MODE = _os.environ.get("HIPAD_LINEAR_MODE", "mfma_bf16")  # "torch_fp32": library fp32 GEMMs (fp32 parity runs)
# whole MLP stacks (Linear / ReLU / LayerNorm sequences) as one forward + two backward launches (hipad_amd.chain)
USE_CHAINS = _os.environ.get("HIPAD_USE_CHAINS", "1") == "1"
LINEAR_INPLACE_GRAD = _os.environ.get("HIPAD_LINEAR_INPLACE_GRAD", "1") == "1"
LINEAR_BWD = _os.environ.get("HIPAD_LINEAR_BWD", "mfma")  # "torch": debugging aid, backward by library matmuls
# ids of the parameters whose gradient some kernel here has accumulated IN PLACE into ``param.grad`` (Linear /
# LayerNorm weights and biases): hipad_amd.dist.FlatGrads uses it to tell them from the parameters autograd
# accumulates itself (see FlatGrads.loosen)
INPLACE_PARAMS = set()
# Every time a GPU tensor is handed to a torch / library implementation instead of a kernel of this package it is
# counted here by reason -- in the default (mfma_bf16) mode a training frame must leave this empty
# (tests/test_graph_step_gpu.py); HIPAD_STRICT=1 turns such a call into an error.  CPU tensors (host-logic tests, the
# oracle's CPU frame) are not kernels' business and are not counted.
import collections as _collections
LIBRARY_CALLS = _collections.Counter()
STRICT = _os.environ.get("HIPAD_STRICT", "0") == "1"


def _library(reason):
    LIBRARY_CALLS[reason] += 1
    if STRICT:
        raise RuntimeError("hipad_amd: %s would run on a torch / library kernel (HIPAD_STRICT=1)" % reason)


class linear_mode:
    """Context manager: with linear_mode("torch_fp32"): ..."""

    def __init__(self, mode):
        if mode not in ("mfma_bf16", "torch_fp32"):
            raise ValueError(mode)
        self.mode = mode

    def __enter__(self):
        global LINEAR_MODE
        self.prev, LINEAR_MODE = LINEAR_MODE, self.mode

    def __exit__(self, *exc):
        global LINEAR_MODE
        LINEAR_MODE = self.prev


class _Linear(Function):
    """y = relu?(x W[r0:r1]^T + b[r0:r1]); weight / bias gradients are accumulated IN PLACE into
    ``weight.grad`` / ``bias.grad`` when those exist (the flat gradient buffer of hipad_amd.dist), so
    the autograd engine has nothing to accumulate for them."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, r0, r1):
        shape = x.shape
        x2 = _c32(x.reshape(-1, shape[-1]))
        w = weight.detach()
        if w.dtype != torch.float32 or not w.is_contiguous():
            raise _lib.HipadError("linear: weight must be a contiguous fp32 tensor")
        wv = w if (r0 == 0 and r1 == w.shape[0]) else w[r0:r1]
        bv = None if bias is None else bias.detach()[r0:r1]
        y = _lib.linear_forward(x2, wv, bv, relu)
        ctx.save_for_backward(x2, y if relu else None)
        ctx.weight, ctx.bias, ctx.rows, ctx.in_shape = weight, bias, (r0, r1), shape
        ctx.mark_non_differentiable()
        return y.view(*shape[:-1], r1 - r0)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2, y_relu = ctx.saved_tensors
        weight, bias, (r0, r1) = ctx.weight, ctx.bias, ctx.rows
        dy2 = _c32(dy.reshape(-1, r1 - r0))
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        need_b = bias is not None and ctx.needs_input_grad[2]
        dx = torch.empty_like(x2) if need_x else None
        ret_w = ret_b = None
        dw = db = None
        if need_w:
            g = weight.grad if LINEAR_INPLACE_GRAD else None
            if g is not None and g.is_contiguous() and g.dtype == torch.float32:
                dw = g[r0:r1]
                INPLACE_PARAMS.add(id(weight))
            else:
                ret_w = torch.zeros_like(weight)
                dw = ret_w[r0:r1]
        if need_b:
            g = bias.grad if LINEAR_INPLACE_GRAD else None
            if g is not None and g.is_contiguous() and g.dtype == torch.float32:
                db = g[r0:r1]
                INPLACE_PARAMS.add(id(bias))
            else:
                ret_b = torch.zeros_like(bias)
                db = ret_b[r0:r1]
        if need_b and not need_w:
            raise _lib.HipadError("linear: bias gradient without weight gradient is not supported")
        wv = weight.detach()[r0:r1]
        if LINEAR_BWD == "torch":
            g = dy2 if y_relu is None else dy2 * (y_relu > 0)
            if dx is not None:
                torch.matmul(g, wv, out=dx)
            if dw is not None:
                dw.add_(g.t() @ x2)
            if db is not None:
                db.add_(g.sum(0))
        else:
            _lib.linear_backward(dy2, y_relu, x2, wv, dx, dw, db)
        return (dx.view(ctx.in_shape) if dx is not None else None), ret_w, ret_b, None, None, None


def linear(x, weight, bias=None, relu=False, rows=None):
    """Linear layer on the MFMA kernel (see include/hipad.h).  ``rows=(r0, r1)`` applies only those
    output rows of ``weight`` / ``bias`` (packed projections) while gradients still land in the full
    parameter's gradient buffer."""
    r0, r1 = (0, weight.shape[0]) if rows is None else rows
    if not (x.is_cuda and LINEAR_MODE == "mfma_bf16"):
        if x.is_cuda:
            _library("linear (mode %s)" % LINEAR_MODE)
        y = torch.nn.functional.linear(x, weight[r0:r1], None if bias is None else bias[r0:r1])
        return torch.relu(y) if relu else y
    return _Linear.apply(x, weight, bias, bool(relu), int(r0), int(r1))


class _LayerNorm(Function):
    """LayerNorm over the last dim; gamma / beta gradients are accumulated IN PLACE into ``weight.grad`` /
    ``bias.grad`` when those exist (as _Linear does), so autograd has nothing to add for them."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        shape = x.shape
        x2 = _c32(x.reshape(-1, shape[-1]))
        need = any(ctx.needs_input_grad[:3])
        y, mean, rstd = _lib.layernorm_forward(x2, None if weight is None else weight.detach(),
                                               None if bias is None else bias.detach(), eps, need_stats=need)
        if need:
            ctx.save_for_backward(x2, mean, rstd)
        ctx.weight, ctx.bias, ctx.in_shape = weight, bias, shape
        return y.view(shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2, mean, rstd = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        dy2 = _c32(dy.reshape(x2.shape))
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        ret_w = ret_b = dgamma = dbeta = None
        if weight is not None and ctx.needs_input_grad[1]:
            g = weight.grad if LINEAR_INPLACE_GRAD else None
            if g is not None and g.is_contiguous() and g.dtype == torch.float32:
                dgamma = g
                INPLACE_PARAMS.add(id(weight))
            else:
                dgamma = ret_w = torch.zeros_like(weight)
        if bias is not None and ctx.needs_input_grad[2]:
            g = bias.grad if LINEAR_INPLACE_GRAD else None
            if g is not None and g.is_contiguous() and g.dtype == torch.float32:
                dbeta = g
                INPLACE_PARAMS.add(id(bias))
            else:
                dbeta = ret_b = torch.zeros_like(bias)
        _lib.layernorm_backward(dy2, x2, mean, rstd, None if weight is None else weight.detach(), dx, dgamma, dbeta)
        return (dx.view(ctx.in_shape) if dx is not None else None), ret_w, ret_b, None


def layer_norm(x, weight, bias, eps=1e-5):
    """LayerNorm over the last dimension on the HIP kernel (see include/hipad.h); CPU tensors (host-logic
    tests) and shapes off the kernel's range go through torch."""
    n = x.shape[-1]
    if not x.is_cuda or n % 4 or n > 1024 or x.dtype != torch.float32:
        if x.is_cuda:
            _library("layer_norm (n = %d, %s)" % (n, x.dtype))
        return torch.nn.functional.layer_norm(x, (n,), weight, bias, eps)
    return _LayerNorm.apply(x, weight, bias, float(eps))


class _LinearReluLN(Function):
    """LayerNorm(relu(x W^T + b)) in one forward launch; backward = the LayerNorm kernel + the fused Linear backward,
    parameter gradients accumulated in place like _Linear / _LayerNorm."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, eps):
        shape = x.shape
        x2 = _c32(x.reshape(-1, shape[-1]))
        y, xr, mean, rstd = _lib.linear_relu_ln_forward(x2, weight.detach(), None if bias is None else bias.detach(),
                                                        gamma.detach(), beta.detach(), eps)
        ctx.save_for_backward(x2, xr, mean, rstd)
        ctx.params, ctx.in_shape = (weight, bias, gamma, beta), shape
        return y.view(*shape[:-1], weight.shape[0])

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x2, xr, mean, rstd = ctx.saved_tensors
        weight, bias, gamma, beta = ctx.params
        dy2 = _c32(dy.reshape(xr.shape))
        rets = [None] * 4

        def target(i, p):
            g = p.grad if LINEAR_INPLACE_GRAD else None
            if g is not None and g.is_contiguous() and g.dtype == torch.float32:
                INPLACE_PARAMS.add(id(p))
                return g
            rets[i] = torch.zeros_like(p)
            return rets[i]

        dxr = torch.empty_like(xr)
        _lib.layernorm_backward(dy2, xr, mean, rstd, gamma.detach(), dxr, target(2, gamma), target(3, beta))
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        _lib.linear_backward(dxr, xr, x2, weight.detach(), dx, target(0, weight), None if bias is None else target(1, bias))
        return (dx.view(ctx.in_shape) if dx is not None else None), rets[0], rets[1], rets[2], rets[3], None


def linear_relu_ln(x, weight, bias, gamma, beta, eps=1e-5):
    """[Linear, ReLU, LayerNorm] unit on one forward kernel; callers check ``linear_relu_ln_ok`` first."""
    return _LinearReluLN.apply(x, weight, bias, gamma, beta, float(eps))


# Measured in the captured training step (MI355X): with the fused unit 52.8 ms per frame, without 50.5 ms -- a
# workgroup that owns full rows (32 x N) re-reads the whole weight matrix and leaves the chip emptier (29 workgroups
# for M = 900) than the 64 x 32-tile GEMM (120) followed by the one-wave-per-row LayerNorm.  Kept as an option
# (HIPAD_FUSE_LINEAR_LN=1) with its parity tests; off by default.
FUSE_LINEAR_LN = _os.environ.get("HIPAD_FUSE_LINEAR_LN", "0") == "1"


def linear_relu_ln_ok(x, weight, gamma):
    """Shapes / layouts the one-launch unit supports (MLPStack additionally requires FUSE_LINEAR_LN)."""
    if not (x.is_cuda and LINEAR_MODE == "mfma_bf16" and x.dtype == torch.float32 and gamma is not None):
        return False
    n, k = weight.shape
    return 16 <= n <= 256 and n % 16 == 0 and k % 4 == 0 and weight.data_ptr() % 16 == 0


class _BoxPointsProject(Function):
    @staticmethod
    def forward(ctx, anchor, fix_scale, learn, projection_mat, image_wh):
        an, fx, ln = _c32(anchor), _c32(fix_scale), _c32(learn)
        pm, wh = _c32(projection_mat), _c32(image_wh)
        ctx.save_for_backward(an, fx, ln, pm, wh)
        return _lib.box_points_project_forward(an, fx, ln, pm, wh)[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_loc):
        an, fx, ln, pm, wh = ctx.saved_tensors
        g_anchor, g_learn = _lib.box_points_project_backward(_c32(grad_loc), an, fx, ln, pm, wh)
        return g_anchor, None, g_learn, None, None


def box_points_project(anchor, fix_scale, learn, projection_mat, image_wh=None):
    """Box key points (fixed + learnable offsets, yaw rotation, translation) projected into every camera:
    (bs,A,D) anchors, (n_fix,3) scales, (bs,A,n_learn*3) pre-sigmoid logits or None -> loc (bs,A,P,cams,2)."""
    return _BoxPointsProject.apply(anchor, fix_scale, learn, projection_mat, image_wh)


class _LinePointsProject(Function):
    @staticmethod
    def forward(ctx, anchor, offset, heights, projection_mat, image_wh, S, Hn, K):
        an, of, hs, pm, wh = _c32(anchor), _c32(offset), _c32(heights), _c32(projection_mat), _c32(image_wh)
        ctx.save_for_backward(an, of, hs, pm, wh)
        ctx.dims = (S, Hn, K)
        return _lib.line_points_project_forward(an, of, hs, pm, wh, S, Hn, K)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_loc):
        an, of, hs, pm, wh = ctx.saved_tensors
        g_anchor, g_offset = _lib.line_points_project_backward(_c32(grad_loc), an, of, hs, pm, wh, *ctx.dims)
        return g_anchor, g_offset, None, None, None, None, None, None


def line_points_project(anchor, offset, heights, projection_mat, image_wh, num_sample, num_heights, num_learnable):
    """Poly-line key points (sample + learned planar offset, at every height) projected into every camera:
    anchor (bs,A,S*2), offset (bs,A,S*Hn*K*2), heights (Hn,) -> loc (bs,A,S*Hn*K,cams,2)."""
    return _LinePointsProject.apply(anchor, offset, heights, projection_mat, image_wh, int(num_sample), int(num_heights),
                                    int(num_learnable))


class _FocalLoss(Function):
    @staticmethod
    def forward(ctx, logits, target, weight, avg_factor, layers, alpha, gamma):
        x = _c32(logits)
        w = None if weight is None else _c32(weight.to(torch.float32))
        a = None if avg_factor is None else _c32(avg_factor.to(torch.float32).reshape(-1).expand(layers))
        loss, grad = _lib.focal_loss_forward(x, target.contiguous(), w, a, layers, alpha, gamma)
        ctx.save_for_backward(grad)
        ctx.layers = layers
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_loss):
        (grad,) = ctx.saved_tensors
        g = grad.view(ctx.layers, -1) * grad_loss.reshape(ctx.layers, 1)
        return g.view_as(grad), None, None, None, None, None, None


def focal_loss(logits, target, weight=None, avg_factor=None, layers=1, alpha=0.25, gamma=2.0):
    """Per-layer sigmoid focal loss (see include/hipad.h): logits (rows, C), integer targets in [0, C] -> (layers,)."""
    return _FocalLoss.apply(logits, target, weight, avg_factor, int(layers), float(alpha), float(gamma))


class _AddRows(Function):
    @staticmethod
    def forward(ctx, base, *rows):
        ctx.shapes = [tuple(r.shape) for r in rows]
        return _lib.add_rows(_c32(base), [_c32(r) for r in rows])

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        g = _c32(gout)
        gr = None
        if any(ctx.needs_input_grad[1:]):
            gr = _lib.rows_sum(g)                       # one reduction serves every row vector
        return (g,) + tuple(gr.view(s) if need else None for s, need in zip(ctx.shapes, ctx.needs_input_grad[1:]))


def add_rows(base, *rows):
    """``base`` (bs, N, C) plus row vectors (bs, 1, C) / (bs, C) broadcast over the N rows: one launch forward; backward
    hands the output gradient to ``base`` as is and ONE row sum to all the vectors (include/hipad.h: hipad_add_rows)."""
    rows = [r for r in rows if r is not None]
    if not rows:
        return base
    if (not base.is_cuda or base.dim() != 3 or base.shape[-1] % 4 or len(rows) > 3
            or any(r.numel() != base.shape[0] * base.shape[-1] for r in rows)):
        out = base
        for r in rows:
            out = out + r
        return out
    return _AddRows.apply(base, *rows)


class _DepthLoss(Function):
    """Dense-depth heads + loss on the flat pyramid (csrc/depthloss.hip).  Inputs: the pyramid (the alias
    shared_feature_grad returned, so that the feature gradient goes into the frame's shared fp32 buffer), its token,
    focal, then per level (gt, weight, bias)."""

    @staticmethod
    def forward(ctx, feat, token, focal, meta, *tensors):
        cams, geometry, equal_focal, max_depth, loss_weight = meta        # geometry: [(rows_per_cam, row_offset)]
        L = len(geometry)
        gts = [_c32(t).reshape(-1) for t in tensors[:L]]
        ws, bs_ = tensors[L:2 * L], tensors[2 * L:3 * L]
        levels = [(gts[i], ws[i].detach().reshape(-1), bs_[i].detach().reshape(-1), geometry[i][0], geometry[i][1]) for i in range(L)]
        focal = None if focal is None else _c32(focal).reshape(-1)
        loss, coef, pred = _lib.depth_loss_forward(feat, focal, levels, cams, equal_focal, max_depth, loss_weight)
        ctx.holder = getattr(feat, "_hipad_grad_holder", None) if token is not None else None
        ctx.meta, ctx.levels, ctx.params = meta, levels, (ws, bs_)
        ctx.save_for_backward(feat, pred, coef)
        ctx.terms = loss.detach()
        return loss[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        feat, pred, coef = ctx.saved_tensors
        cams, geometry, _, max_depth, _ = ctx.meta
        L = len(geometry)
        ws, bs_ = ctx.params
        grads, rets = [], [None] * (3 * L)
        for i in range(L):
            pair = []
            for k, p in ((L + i, ws[i]), (2 * L + i, bs_[i])):
                tgt = None
                if ctx.needs_input_grad[4 + k]:
                    gp = p.grad if (LINEAR_INPLACE_GRAD and p.is_leaf) else None
                    if gp is not None and gp.is_contiguous() and gp.dtype == torch.float32:
                        INPLACE_PARAMS.add(id(p))
                        tgt = gp
                    else:
                        tgt = rets[k] = torch.zeros_like(p, dtype=torch.float32)
                pair.append(tgt)
            grads.append(tuple(pair))
        grad_token = ret_feat = None
        if ctx.holder is not None:
            # the frame's shared pyramid gradient (the aggregation calls' buffer): created here if this node runs first
            key = torch.cuda.current_stream(feat.device).cuda_stream
            buf = ctx.holder["bufs"].get(key)
            if buf is None:
                buf = ctx.holder["bufs"][key] = torch.zeros(feat.shape, dtype=torch.float32, device=feat.device)
            grad_token = ctx.holder.get("zero")
            if grad_token is None:
                grad_token = torch.zeros((), dtype=feat.dtype, device=feat.device)
        else:
            buf = ret_feat = torch.zeros(feat.shape, dtype=torch.float32, device=feat.device)
        _lib.depth_loss_backward(buf, pred, coef, _c32(g).reshape(-1), feat, ctx.levels, grads, cams, max_depth)
        return (ret_feat, grad_token, None, None) + tuple(rets)


def depth_loss(feat, focal, gts, weights, biases, geometry, cams, equal_focal, max_depth, loss_weight):
    """Masked-L1 dense-depth loss of the 1x1 heads (``weights[l]`` (1, 256, 1, 1), ``biases[l]`` (1,)) on the rows of
    the flat bf16 pyramid ``feat`` (bs, rows, 256); ``geometry[l]`` = (rows per camera, first row of the level inside a
    sample); ``gts[l]``: (bs * cams, h, w).  One launch forward (+ a one-thread finish), one backward."""
    token = getattr(feat, "_hipad_grad_token", None)
    meta = (int(cams), tuple((int(a), int(b)) for a, b in geometry), float(equal_focal), float(max_depth), float(loss_weight))
    return _DepthLoss.apply(feat, token, focal, meta, *gts, *weights, *biases)


class _StepOffsets(Function):
    @staticmethod
    def forward(ctx, x):
        return _lib.step_offsets(_c32(x))

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        return _lib.step_offsets(_c32(gout), adjoint=True)


def step_offsets(x):
    """(..., steps, dims) way-points -> offsets between consecutive steps (the first step keeps its value): one launch
    forward, one backward (include/hipad.h: hipad_step_offsets).  CPU tensors: the torch expression."""
    if not x.is_cuda:
        return torch.cat([x[..., :1, :], x[..., 1:, :] - x[..., :-1, :]], dim=-2)
    return _StepOffsets.apply(x)


class _ChunkMix(Function):
    @staticmethod
    def forward(ctx, x0, x1, table, rows):
        a = _c32(x0)
        b = None if x1 is None else _c32(x1)
        ctx.table, ctx.rows = table, rows
        ctx.two = x1 is not None
        return _lib.chunk_mix(a, b, table, rows)

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        t = ctx.table
        tt = tuple(tuple(t[g][k] for g in range(len(t))) for k in range(len(t[0])))   # transposed
        dx = _lib.chunk_mix(_c32(gout), None, tt, ctx.rows)
        return dx, (dx if ctx.two else None), None, None


def chunk_mix(x0, x1, table, rows):
    """Chunks of ``rows`` rows along dim 1: out chunk g = sum_k table[g][k] * (x0 chunk k + x1 chunk k) in ONE launch (and
    one for the backward); see include/hipad.h.  CPU tensors: the same expression in torch ops."""
    table = tuple(tuple(float(v) for v in row) for row in table)
    if not x0.is_cuda:
        x = x0 if x1 is None else x0 + x1
        chunks = x.split(rows, dim=1)
        outs = []
        for row in table:
            acc = None
            for w, c in zip(row, chunks):
                if w != 0.0:
                    acc = c * w if acc is None else acc + c * w
            outs.append(acc if acc is not None else torch.zeros_like(chunks[0]))
        return torch.cat(outs, dim=1)
    return _ChunkMix.apply(x0, x1, table, int(rows))
