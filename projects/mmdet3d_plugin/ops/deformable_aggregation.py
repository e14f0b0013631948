"""autograd glue for the deformable aggregation op.

Mirrors the interface of the reference's ``DeformableAggregationFunction``
(ops/deformable_aggregation.py:7-75): ``apply(mc_ms_feat, spatial_shape, scale_start_index,
sampling_location, weights) -> output`` with gradients for feat, location and weights.

Differences underneath (same results):
  * int32 copies of the two index tables are cached on the tensors instead of being
    re-cast by a kernel on each of the 24 calls per forward;
  * backward lets the kernel write grad_location / grad_weights outright (no memsets);
  * if the feature tensor came through ``shared_feature_grad`` the 24 call sites add their
    feature gradient into ONE buffer instead of 24x (zeros + autograd add) of the pyramid -- and they do it in ONE
    pass: each call's backward only computes grad_location / grad_weights and leaves its (location, weights,
    grad_output) triple with the sink, whose own backward node (it runs after every call's) hands the whole table to
    ``hipad_daf_backward_feat_multi``: one counting sort + one accumulation for the frame instead of 24 five-launch
    pipelines (``HIPAD_DAF_DEFER=0`` restores the per-call pipelines).
"""
import torch
from torch.autograd.function import Function, once_differentiable

from hipad_amd import lib as _lib


import os as _os

_ATOMIC_FEAT = _os.environ.get("HIPAD_DAF_ATOMIC", "0") == "1"  # debugging aid: one-pass atomic scatter backward
DEFER_FEAT = _os.environ.get("HIPAD_DAF_DEFER", "1") == "1"      # feature gradient of all call sites in one pass (sink)


_CROSS_CHECK = _os.environ.get("HIPAD_DAF_CROSS_CHECK", "0") == "1"  # debugging aid, see _cross_check
CROSS_CHECK_LOG = []  # per backward call site: device tensor [max|sorted-atomic|, max|atomic|, nonfinite flags]


def _cross_check(feat, ss, st, loc, w, grad_output):
    """Debugging aid (capturable): runs the feature gradient of this call through BOTH backward variants into
    scratch buffers and records how far they are apart, plus finiteness of the inputs."""
    a = torch.zeros_like(feat)
    b = torch.zeros_like(feat)
    _lib.daf_backward(feat, ss, st, loc, w, grad_output, a, None, None, atomic_feat=True)
    _lib.daf_backward(feat, ss, st, loc, w, grad_output, b, None, None, atomic_feat=False)
    rec = torch.stack([(a - b).abs().max(), a.abs().max(), (~torch.isfinite(b)).sum().float(),
                       (~torch.isfinite(loc)).sum().float(), (~torch.isfinite(w)).sum().float(),
                       (~torch.isfinite(grad_output)).sum().float()])
    CROSS_CHECK_LOG.append((tuple(loc.shape), rec))


def _as_i32(t):
    if t.dtype == torch.int32 and t.is_contiguous():
        return t
    # int32 twin cached on the tensor object, keyed on (storage address, version counter): a caller that rewrites
    # spatial_shape / scale_start_index in place gets a fresh copy instead of stale row addresses
    key = (t.data_ptr(), t._version)
    cached = getattr(t, "_hipad_i32", None)
    if cached is None or cached[0] != key or cached[1].device != t.device:
        cached = (key, t.contiguous().int())
        try:
            t._hipad_i32 = cached
        except Exception:
            pass
    return cached[1]


def _as_f32(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _feat_rows(t, weights):
    """The feature tensor as the kernels take it: the encoder's bf16 rows as they are when the 256-channel kernels apply
    (hipad_daf_forward_bf16 / _backward_bf16: same values, half the gather bytes), else contiguous fp32."""
    if (t.dtype == torch.bfloat16 and t.is_cuda and t.is_contiguous() and t.shape[-1] == 256 and weights.shape[-1] == 8
            and not _ATOMIC_FEAT and not _CROSS_CHECK):
        return t
    return _as_f32(t)


class _FeatureGradSink(Function):
    """Identity on the feature tensor; collects d(loss)/d(feat) of all consumers in one buffer."""

    @staticmethod
    def forward(ctx, feat):
        out = feat.view_as(feat)
        ctx.set_materialize_grads(False)    # no direct consumer of `out` -> grad_feat arrives as None, not as zeros
        # "bufs": one accumulation buffer per stream that runs aggregation backwards (per-call pipelines);
        # "pending": the (feat, ss, st, loc, w, grad_out) of the call sites whose feature gradient is deferred to this node
        ctx.holder = holder = {"bufs": {}, "pending": []}
        out._hipad_grad_holder = holder
        token = torch.zeros((), dtype=feat.dtype, device=feat.device)
        holder["zero"] = token.detach()     # the (constant) gradient every consumer returns for its token edge
        return out, token

    @staticmethod
    def backward(ctx, grad_feat, grad_token):
        bufs = list(ctx.holder["bufs"].values())
        ctx.holder["bufs"] = {}
        pending, ctx.holder["pending"] = ctx.holder["pending"], []
        if pending:
            feat, ss, st = pending[0][:3]
            buf = bufs[0] if bufs else torch.zeros(feat.shape, dtype=torch.float32, device=feat.device)
            if not bufs:
                bufs = [buf]
            _lib.daf_backward_feat_multi([c[3:] for c in pending], buf, ss, st)
        total = grad_feat
        here = torch.cuda.current_stream(grad_token.device) if (grad_token is not None and grad_token.is_cuda) else None
        for buf in bufs:  # the token edges made the engine order this node after every producer stream
            if here is not None:
                buf.record_stream(here)
            total = buf if total is None else total + buf
        return total


def shared_feature_grad(feat):
    """Route the feature gradients of every later aggregation call into one shared buffer.

    Returns a tensor equal to ``feat``; pass it (inside the usual
    ``[col_feats, spatial_shape, scale_start_index]`` triple) to the aggregation calls.
    """
    if not (feat.requires_grad and torch.is_grad_enabled()):
        return feat
    out, token = _FeatureGradSink.apply(feat)
    out._hipad_grad_token = token
    out._hipad_grad_holder = out._hipad_grad_holder  # keep attribute on the returned alias
    return out


class DeformableAggregationFunction(Function):
    @staticmethod
    def forward(ctx, mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights, token=None):
        holder = getattr(mc_ms_feat, "_hipad_grad_holder", None)
        feat = _feat_rows(mc_ms_feat, weights)
        ss = _as_i32(spatial_shape)
        st = _as_i32(scale_start_index)
        loc = _as_f32(sampling_location)
        w = _as_f32(weights)
        output = _lib.daf_forward(feat, ss, st, loc, w)
        ctx.save_for_backward(feat, ss, st, loc, w)
        ctx.holder = holder if token is not None else None
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        feat, ss, st, loc, w = ctx.saved_tensors
        need_feat, _, _, need_loc, need_w = ctx.needs_input_grad[:5]
        grad_output = _as_f32(grad_output)
        grad_loc = torch.empty_like(loc) if need_loc else None
        grad_w = torch.empty_like(w) if need_w else None
        grad_feat = ret_feat = None
        grad_token = None
        deferrable = (DEFER_FEAT and not _ATOMIC_FEAT and not _CROSS_CHECK and feat.is_cuda and feat.shape[-1] == 256
                      and w.shape[-1] == 8 and ss.shape[1] <= 8)
        if ctx.holder is not None and deferrable:
            # the sink's backward runs the feature gradient of ALL call sites in one pass (see the module docstring); the
            # token edge orders that node behind this one, on whatever stream this one ran
            ctx.holder["pending"].append((feat, ss, st, loc, w, grad_output))
            grad_token = ctx.holder.get("zero")
            if grad_token is None:
                grad_token = torch.zeros((), dtype=feat.dtype, device=feat.device)
        elif ctx.holder is not None:
            # shared sink: every call site running on THIS stream adds into the same buffer (calls on one stream
            # are serialised; concurrent streams get a buffer each: the kernel's read-modify-write of rows is
            # exclusive only within a launch); the sink node sums the buffers
            key = torch.cuda.current_stream(feat.device).cuda_stream
            grad_feat = ctx.holder["bufs"].get(key)
            if grad_feat is None:
                grad_feat = ctx.holder["bufs"][key] = torch.zeros(feat.shape, dtype=torch.float32, device=feat.device)
            grad_token = ctx.holder.get("zero")
            if grad_token is None:
                grad_token = torch.zeros((), dtype=feat.dtype, device=feat.device)
        elif need_feat:
            grad_feat = ret_feat = torch.zeros(feat.shape, dtype=torch.float32, device=feat.device)
        if _CROSS_CHECK and grad_feat is not None:
            _cross_check(feat, ss, st, loc, w, grad_output)
        _lib.daf_backward(feat, ss, st, loc, w, grad_output, grad_feat, grad_loc, grad_w, overwrite_loc_w=True,
                          atomic_feat=_ATOMIC_FEAT)
        return ret_feat, None, None, grad_loc, grad_w, grad_token
