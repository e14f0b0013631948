"""MLP chains: whole `linear_relu_ln` stacks / heads of the decoder as ONE forward and TWO backward launches
(hip-ad_amd/csrc/chain.hip, include/hipad.h "MLP chains"), for one chain or a group of independent chains.

Host-side plumbing only: parses a ``nn.Sequential`` of the reference's building blocks (reference
models/blocks.py:32-42 ``linear_relu_ln``; heads in det/map/plan/motion/ego ``blocks.py``) into a ``ChainSpec``,
keeps bf16 (+ transposed bf16) copies of the weights, fills the C descriptor structs and wraps the three launches in
one autograd Function.  Parameter gradients are accumulated IN PLACE into ``param.grad`` where that exists (the flat
gradient buffer of hipad_amd.dist), like functional._Linear / _LayerNorm do.
"""
import ctypes

import torch
import torch.nn as nn
from torch.autograd.function import Function, once_differentiable

from . import lib as _lib

MAX_LAYERS, MAX_CHAINS, MAX_DW, NONE = 6, 8, 48, 0xFFFFFFFF
MAX_WIDTH = 256

c_f = ctypes.c_float
c_p = ctypes.c_void_p


class CLayer(ctypes.Structure):
    _fields_ = [("w", c_p), ("bias", c_p), ("gamma", c_p), ("beta", c_p), ("off_h", ctypes.c_uint), ("off_y", ctypes.c_uint),
                ("off_stats", ctypes.c_uint), ("K", ctypes.c_int), ("N", ctypes.c_int), ("flags", ctypes.c_int), ("eps", c_f)]


class CChain(ctypes.Structure):
    _fields_ = [("x0", c_p), ("x1", c_p), ("xsum", c_p), ("out", c_p), ("out_scale", c_p), ("residual", c_p), ("save", c_p),
                ("ldx0", ctypes.c_int), ("ldx1", ctypes.c_int), ("ldo", ctypes.c_int), ("ldr", ctypes.c_int),
                ("M", ctypes.c_int), ("nlayers", ctypes.c_int), ("layers", CLayer * MAX_LAYERS)]


class CGradLayer(ctypes.Structure):
    _fields_ = [("wt", c_p), ("gamma", c_p), ("dgamma", c_p), ("dbeta", c_p), ("off_h", ctypes.c_uint),
                ("off_stats", ctypes.c_uint), ("off_dy", ctypes.c_uint), ("K", ctypes.c_int), ("N", ctypes.c_int),
                ("flags", ctypes.c_int), ("eps", c_f)]


class CChainGrad(ctypes.Structure):
    _fields_ = [("dout", c_p), ("out_scale", c_p), ("dscale", c_p), ("dx", c_p), ("save", c_p), ("dy", c_p),
                ("ldo", ctypes.c_int), ("lddx", ctypes.c_int), ("M", ctypes.c_int), ("nlayers", ctypes.c_int),
                ("layers", CGradLayer * MAX_LAYERS)]


class CDw(ctypes.Structure):
    _fields_ = [("dy", c_p), ("x", c_p), ("dw", c_p), ("db", c_p), ("M", ctypes.c_int), ("N", ctypes.c_int),
                ("K", ctypes.c_int), ("ldx", ctypes.c_int)]


# ------------------------------------------------------------------------------------------------------------
# bf16 copies of the weights
# ------------------------------------------------------------------------------------------------------------
def pack_fragments(a):
    """P(A) of include/hipad.h (hipad_pack_weights): the bf16 copy of a 2-D matrix in MFMA-fragment order, zero-padded to
    16-row x 32-column blocks; block (tr, s), lane = 16 quad + l15 owns A[16 tr + l15][32 s + 8 quad + 0..7]."""
    r, d = a.shape
    tr, s = (r + 15) // 16, (d + 31) // 32
    pad = torch.zeros(tr * 16, s * 32, dtype=torch.float32, device=a.device)
    pad[:r, :d] = a
    return pad.view(tr, 16, s, 4, 8).permute(0, 2, 3, 1, 4).contiguous().to(torch.bfloat16).view(-1)


def packed_numel(rows, depth):
    return ((rows + 15) // 16) * ((depth + 31) // 32) * 512


def bf16_pair(weight):
    """(P(W), P(W^T)): the chain kernels' bf16 operand copies of a 2-D fp32 parameter W [N][K].

    An optimiser that keeps them current (hipad_amd.optim.FlatAdamW: one pack launch per step) attaches them as
    ``weight._hipad_shadow``; otherwise the pair is derived here and cached on (storage address, version counter), so
    in-place torch updates are seen."""
    pair = getattr(weight, "_hipad_shadow", None)
    if pair is not None:
        return pair
    # cached ON the parameter object (a table keyed by id() would hand a new parameter that reuses a freed one's id,
    # address, version and shape the old one's copies)
    hit = getattr(weight, "_hipad_pair", None)
    stamp = (weight.data_ptr(), weight._version, tuple(weight.shape))
    if hit is None or hit[0] != stamp:
        w = weight.detach().float()
        hit = (stamp, pack_fragments(w), pack_fragments(w.t()))
        weight._hipad_pair = hit
    return hit[1], hit[2]


# ------------------------------------------------------------------------------------------------------------
# chain specification
# ------------------------------------------------------------------------------------------------------------
class _L:
    __slots__ = ("weight", "bias", "relu", "ln")

    def __init__(self, lin):
        self.weight, self.bias, self.relu, self.ln = lin.weight, lin.bias, False, None


class ChainSpec:
    """Layers of one chain: [(weight, bias, relu, LayerNorm module or None)], optional trailing Scale parameter."""

    def __init__(self, layers, scale=None):
        self.layers, self.scale = layers, scale
        self.K0, self.N_out = layers[0].weight.shape[1], layers[-1].weight.shape[0]

    def params(self):
        ps = []
        for L in self.layers:
            ps += [L.weight, L.bias]
            ps += [L.ln.weight, L.ln.bias] if L.ln is not None else [None, None]
        ps.append(self.scale)
        return ps


def spec_of(seq):
    """ChainSpec of an ``nn.Sequential`` made of Linear / (Fused)ReLU / LayerNorm / Scale / Identity / Dropout(0)
    modules, or None when the stack does not fit the kernel (cached on the module)."""
    cached = getattr(seq, "_hipad_chain_spec", 0)
    if cached != 0:
        return cached
    from .compat import FusedReLU, Scale
    layers, scale, ok = [], None, True
    for m in seq:
        if isinstance(m, nn.Linear):
            if scale is not None:
                ok = False
            layers.append(_L(m))
            layers[-1].relu = bool(getattr(m, "fuse_relu", False))
        elif isinstance(m, (FusedReLU, nn.ReLU)):
            if not layers or layers[-1].ln is not None:
                ok = False
            else:
                layers[-1].relu = True
        elif isinstance(m, nn.LayerNorm):
            if (not layers or layers[-1].ln is not None or len(m.normalized_shape) != 1
                    or m.normalized_shape[0] != layers[-1].weight.shape[0] or m.weight is None):
                ok = False
            else:
                layers[-1].ln = m
        elif isinstance(m, Scale):
            if not layers or scale is not None or layers[-1].ln is not None or m.scale.dim() != 1:
                ok = False
            scale = m.scale
        elif isinstance(m, nn.Identity) or (isinstance(m, nn.Dropout) and m.p == 0.0):
            pass
        else:
            ok = False
        if not ok:
            break
    if ok and (not layers or len(layers) > MAX_LAYERS):
        ok = False
    if ok:
        for L in layers:
            n, k = L.weight.shape
            if n > MAX_WIDTH or k > MAX_WIDTH or L.weight.dtype != torch.float32:
                ok = False
        if scale is not None and scale.numel() != layers[-1].weight.shape[0]:
            ok = False
    spec = ChainSpec(layers, scale) if ok else None
    try:
        seq._hipad_chain_spec = spec
    except Exception:  # noqa: BLE001
        pass
    return spec


class Call:
    """One chain invocation: out = spec(x0 (+ x1)) (+ residual).  ``out_slot`` = (index of a shared output tensor,
    first column) when several calls write column ranges of one tensor (concatenated encoders)."""

    __slots__ = ("spec", "x0", "x1", "residual", "out_slot", "out_width")

    def __init__(self, spec, x0, x1=None, residual=None, out_slot=None, out_width=None):
        self.spec, self.x0, self.x1, self.residual, self.out_slot, self.out_width = spec, x0, x1, residual, out_slot, out_width


def _rows_view(t, width):
    """(tensor usable as M rows of `width` floats with a uniform row stride, M, ld)."""
    if t.dtype != torch.float32:
        t = t.float()
    if t.shape[-1] != width:
        raise _lib.HipadError(f"chain: input width {t.shape[-1]} != layer width {width}")
    ok = t.dim() >= 1 and t.stride(-1) == 1
    if ok and t.dim() >= 2:
        ld = t.stride(-2) if t.shape[-2] > 1 else max(width, 1)
        for d in range(t.dim() - 3, -1, -1):
            if t.shape[d] != 1 and t.stride(d) != t.stride(d + 1) * t.shape[d + 1]:
                ok = False
        if t.dim() >= 3 and t.shape[-2] == 1:
            ok = False  # degenerate row dim: let contiguous() sort it out
        if ld < width:
            ok = False
    elif ok:
        ld = width
    if not ok:
        t = t.contiguous()
        ld = width
    M = t.numel() // width
    return t, M, ld


def _acc_target(p, rets, i):
    """In-place accumulation target for parameter ``p`` (its .grad when that is a contiguous fp32 buffer) or a fresh
    zero tensor that is returned to autograd instead."""
    from . import functional as HF
    g = p.grad if HF.LINEAR_INPLACE_GRAD else None
    if g is not None and g.is_contiguous() and g.dtype == torch.float32:
        HF.INPLACE_PARAMS.add(id(p))
        return g
    rets[i] = torch.zeros_like(p)
    return rets[i]


class _Chains(Function):
    @staticmethod
    def forward(ctx, meta, *tensors):
        calls, n_out = meta          # calls: list of (spec, out_slot, out_width); tensors: x0, x1, residual, params ...
        lib = _lib.load()
        dev = tensors[0].device
        need_grad = any(ctx.needs_input_grad[1:])
        arr = (CChain * len(calls))()
        outs = [None] * n_out
        keep = []          # python references that must outlive the launch / the backward
        rec = []           # per call: dict for backward
        pos = 0
        for ci, (spec, out_slot, out_width) in enumerate(calls):
            x0, x1, res = tensors[pos:pos + 3]
            nparam = 4 * len(spec.layers) + 1
            params = tensors[pos + 3:pos + 3 + nparam]
            pin = pos
            pos += 3 + nparam
            K0 = spec.K0
            x0v, M, ld0 = _rows_view(x0, K0)
            x1v = ld1 = None
            if x1 is not None:
                x1v, M1, ld1 = _rows_view(x1, K0)
                if M1 != M:
                    raise _lib.HipadError("chain: x0 and x1 differ in rows")
            N_out = spec.N_out
            if out_slot is None:
                out = torch.empty(M, N_out, dtype=torch.float32, device=dev)
                oslot, ocol, ldo = ci, 0, N_out
                outs[ci] = out
                out_ptr = out.data_ptr()
            else:
                oslot, ocol = out_slot
                if outs[oslot] is None:
                    outs[oslot] = torch.empty(M, out_width, dtype=torch.float32, device=dev)
                out = outs[oslot]
                ldo = out.shape[1]
                out_ptr = out.data_ptr() + 4 * ocol
            resv = ldr = None
            if res is not None:
                resv, Mr, ldr = _rows_view(res, N_out)
                if Mr != M:
                    raise _lib.HipadError("chain: residual differs in rows")
            # saved activations
            offs, total = [], 0
            nl = len(spec.layers)
            for li, L in enumerate(spec.layers):
                N = L.weight.shape[0]
                last = li + 1 == nl
                need_h = need_grad and ((not last) or L.relu or L.ln is not None or spec.scale is not None)
                oh = oy = ost = NONE
                if need_h:
                    oh, total = total, total + M * N
                if need_grad and L.ln is not None:
                    oy, total = total, total + M * N
                    ost, total = total, total + 2 * M
                offs.append((oh, oy, ost))
            save = torch.empty(max(total, 1), dtype=torch.float32, device=dev) if need_grad else None
            xsum = torch.empty(M, K0, dtype=torch.float32, device=dev) if (need_grad and x1 is not None) else None
            c = arr[ci]
            c.x0, c.x1 = x0v.data_ptr(), (x1v.data_ptr() if x1v is not None else None)
            c.xsum = xsum.data_ptr() if xsum is not None else None
            c.out = out_ptr
            scale = params[-1]
            c.out_scale = scale.data_ptr() if scale is not None else None
            c.residual = resv.data_ptr() if resv is not None else None
            c.save = save.data_ptr() if save is not None else None
            c.ldx0, c.ldx1, c.ldo, c.ldr = ld0, (ld1 or 0), ldo, (ldr or 0)
            c.M, c.nlayers = M, nl
            shadows = []
            for li, L in enumerate(spec.layers):
                w, b, g, bt = params[4 * li:4 * li + 4]
                wb, wtb = bf16_pair(L.weight)
                shadows.append((wb, wtb))
                cl = c.layers[li]
                cl.w = wb.data_ptr()
                cl.bias = b.data_ptr() if b is not None else None
                cl.gamma = g.data_ptr() if g is not None else None
                cl.beta = bt.data_ptr() if bt is not None else None
                cl.off_h, cl.off_y, cl.off_stats = offs[li]
                cl.N, cl.K = L.weight.shape
                cl.flags = (1 if L.relu else 0) | (2 if L.ln is not None else 0)
                cl.eps = float(L.ln.eps) if L.ln is not None else 0.0
            keep.append((x0v, x1v, resv, shadows))
            rec.append(dict(spec=spec, M=M, ld0=ld0, x0v=x0v, xsum=xsum, save=save, offs=offs, oslot=oslot, ocol=ocol,
                            pin=pin, x0_shape=tuple(x0.shape), x1_shape=None if x1 is None else tuple(x1.shape),
                            res_shape=None if res is None else tuple(res.shape), shadows=shadows))
        with torch.cuda.device(dev):
            _lib.check(lib.hipad_chain_forward(arr, len(calls), _lib.stream_ptr(dev)), "hipad_chain_forward")
        ctx.rec, ctx.n_in = rec, len(tensors)
        ctx.keep = keep
        results = []
        for ci, (spec, out_slot, out_width) in enumerate(calls):
            if out_slot is None:
                r = rec[ci]
                results.append(outs[ci].view(*r["x0_shape"][:-1], spec.N_out))
        # shared (concatenated) outputs come after the private ones, in slot order
        shared = sorted({r["oslot"] for r, (s, o, w) in zip(rec, calls) if o is not None})
        for slot in shared:
            r = next(r for r, (s, o, w) in zip(rec, calls) if o is not None and r["oslot"] == slot)
            results.append(outs[slot].view(*r["x0_shape"][:-1], outs[slot].shape[1]))
        ctx.order = ([ci for ci, (s, o, w) in enumerate(calls) if o is None], shared)
        ctx.calls = calls
        return tuple(results)

    @staticmethod
    @once_differentiable
    def backward(ctx, *gouts):
        lib = _lib.load()
        calls, rec = ctx.calls, ctx.rec
        private, shared = ctx.order
        dev = gouts[0].device if gouts[0] is not None else rec[0]["x0v"].device
        grads = [None] * (ctx.n_in + 1)      # +1: meta
        gmap = {}
        for k, ci in enumerate(private):
            gmap[("p", ci)] = gouts[k]
        for k, slot in enumerate(shared):
            gmap[("s", slot)] = gouts[len(private) + k]
        garr = (CChainGrad * len(calls))()
        dws = []
        keep = []
        for ci, ((spec, out_slot, out_width), r) in enumerate(zip(calls, rec)):
            M, nl = r["M"], len(spec.layers)
            N_out = spec.N_out
            if out_slot is None:
                g = gmap[("p", ci)]
                if g is None:
                    g = torch.zeros(M, N_out, dtype=torch.float32, device=dev)
                g2 = g.reshape(M, N_out)
                if g2.dtype != torch.float32 or g2.stride(-1) != 1 or g2.stride(0) < N_out:
                    g2 = g2.float().contiguous()
                dout_ptr, ldo = g2.data_ptr(), g2.stride(0) if M > 1 else N_out
            else:
                g = gmap[("s", r["oslot"])]
                width = out_width
                if g is None:
                    g = torch.zeros(M, width, dtype=torch.float32, device=dev)
                g2 = g.reshape(M, width)
                if g2.dtype != torch.float32 or not g2.is_contiguous():
                    g2 = g2.float().contiguous()
                dout_ptr, ldo = g2.data_ptr() + 4 * r["ocol"], width
            keep.append(g2)
            pin = r["pin"]
            need_x = ctx.needs_input_grad[1 + pin] or (r["x1_shape"] is not None and ctx.needs_input_grad[2 + pin])
            dy_total = sum(M * L.weight.shape[0] for L in spec.layers)
            dy = torch.empty(dy_total, dtype=torch.float32, device=dev)
            dx = torch.empty(M, spec.K0, dtype=torch.float32, device=dev) if need_x else None
            rets = [None] * (4 * nl + 1)
            c = garr[ci]
            c.dout = dout_ptr
            c.out_scale = spec.scale.data_ptr() if spec.scale is not None else None
            c.dscale = None
            if spec.scale is not None and ctx.needs_input_grad[1 + pin + 3 + 4 * nl]:
                c.dscale = _acc_target(spec.scale, rets, 4 * nl).data_ptr()
            c.dx = dx.data_ptr() if dx is not None else None
            c.save = r["save"].data_ptr()
            c.dy = dy.data_ptr()
            c.ldo, c.lddx, c.M, c.nlayers = ldo, spec.K0, M, nl
            off = 0
            for li, L in enumerate(spec.layers):
                N, K = L.weight.shape
                cl = c.layers[li]
                cl.wt = r["shadows"][li][1].data_ptr()
                oh, oy, ost = r["offs"][li]
                cl.off_h, cl.off_stats, cl.off_dy = oh, ost, off
                cl.K, cl.N = K, N
                cl.flags = (1 if L.relu else 0) | (2 if L.ln is not None else 0)
                cl.eps = float(L.ln.eps) if L.ln is not None else 0.0
                cl.gamma = cl.dgamma = cl.dbeta = None
                if L.ln is not None:
                    cl.gamma = L.ln.weight.data_ptr()
                    if ctx.needs_input_grad[1 + pin + 3 + 4 * li + 2]:
                        cl.dgamma = _acc_target(L.ln.weight, rets, 4 * li + 2).data_ptr()
                    if L.ln.bias is not None and ctx.needs_input_grad[1 + pin + 3 + 4 * li + 3]:
                        cl.dbeta = _acc_target(L.ln.bias, rets, 4 * li + 3).data_ptr()
                if ctx.needs_input_grad[1 + pin + 3 + 4 * li]:
                    d = CDw()
                    d.dy = dy.data_ptr() + 4 * off
                    if li == 0:
                        xs = r["xsum"] if r["xsum"] is not None else r["x0v"]
                        d.x, d.ldx = xs.data_ptr(), (spec.K0 if r["xsum"] is not None else r["ld0"])
                    else:
                        poh, poy, _ = r["offs"][li - 1]
                        d.x = r["save"].data_ptr() + 4 * (poy if spec.layers[li - 1].ln is not None else poh)
                        d.ldx = K
                    d.dw = _acc_target(L.weight, rets, 4 * li).data_ptr()
                    d.db = None
                    if L.bias is not None and ctx.needs_input_grad[1 + pin + 3 + 4 * li + 1]:
                        d.db = _acc_target(L.bias, rets, 4 * li + 1).data_ptr()
                    d.M, d.N, d.K = M, N, K
                    dws.append(d)
                off += M * N
            keep.append((dy, dx, rets))
            # gradients of this call's inputs
            if dx is not None:
                if ctx.needs_input_grad[1 + pin]:
                    grads[1 + pin] = dx.view(r["x0_shape"])
                if r["x1_shape"] is not None and ctx.needs_input_grad[2 + pin]:
                    grads[2 + pin] = dx.view(r["x1_shape"])
            if r["res_shape"] is not None and ctx.needs_input_grad[3 + pin]:
                gr = g2[:, r["ocol"]:r["ocol"] + N_out] if out_slot is not None else g2
                grads[3 + pin] = gr.reshape(r["res_shape"])
            for k, t in enumerate(rets):
                if t is not None:
                    grads[1 + pin + 3 + k] = t
        with torch.cuda.device(dev):
            st = _lib.stream_ptr(dev)
            _lib.check(lib.hipad_chain_backward_dx(garr, len(calls), st), "hipad_chain_backward_dx")
            if dws:
                darr = (CDw * len(dws))(*dws)
                _lib.check(lib.hipad_chain_backward_dw(darr, len(dws), st), "hipad_chain_backward_dw")
        return tuple(grads)


def run(calls):
    """Run a group of independent chains in one forward launch (two backward launches).

    ``calls``: list of ``Call``.  Returns the private outputs (calls without ``out_slot``) in call order, followed by the
    shared output tensors in slot order.  Every input must be a CUDA fp32 tensor."""
    flat, meta = [], []
    slots = sorted({c.out_slot[0] for c in calls if c.out_slot is not None})
    remap = {s: len(calls) + i for i, s in enumerate(slots)}   # shared outputs live after the private slots
    for c in calls:
        flat += [c.x0, c.x1, c.residual] + c.spec.params()
        slot = None if c.out_slot is None else (remap[c.out_slot[0]], c.out_slot[1])
        meta.append((c.spec, slot, c.out_width))
    return _Chains.apply((meta, len(calls) + len(slots)), *flat)


def usable(x):
    from . import functional as HF
    return x.is_cuda and HF.LINEAR_MODE == "mfma_bf16" and HF.USE_CHAINS
