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


def _shadow_ok(weight):
    """The optimiser-kept copies still describe the parameter (hipad_amd.optim.shadow_is_current: a torch write since the
    last refresh -- load_state_dict, checkpoint resume -- triggers a refresh, or sends the caller to its own cache)."""
    from .optim import shadow_is_current
    return shadow_is_current(weight)


def bf16_pair(weight):
    """(P(W), P(W^T)): the chain kernels' bf16 operand copies of a 2-D fp32 parameter W [N][K].

    An optimiser that keeps them current (hipad_amd.optim.FlatAdamW: one pack launch per step) attaches them as
    ``weight._hipad_shadow``; otherwise the pair is derived here and cached on (storage address, version counter), so
    in-place torch updates are seen."""
    pair = getattr(weight, "_hipad_shadow", None)
    if pair is not None and _shadow_ok(weight):
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
    """One Linear of a chain: the whole ``weight`` [N][K] / ``bias`` [N] of a module, or -- ``rows`` = (r0, r1) -- a row
    block of a packed projection (attention in_proj_weight), which is a contiguous [r1 - r0][K] matrix of its own."""
    __slots__ = ("weight", "bias", "relu", "ln", "rows", "N", "K")

    def __init__(self, lin=None, weight=None, bias=None, rows=None):
        self.weight, self.bias = (lin.weight, lin.bias) if lin is not None else (weight, bias)
        self.relu, self.ln, self.rows = False, None, rows
        self.N = self.weight.shape[0] if rows is None else rows[1] - rows[0]
        self.K = self.weight.shape[1]

    @property
    def r0(self):
        return 0 if self.rows is None else self.rows[0]

    def operands(self):
        """(P(W), P(W^T)) of this layer's matrix."""
        if self.rows is None:
            return bf16_pair(self.weight)
        table = getattr(self.weight, "_hipad_shadow_rows", None)
        if table is not None and self.rows in table and _shadow_ok(self.weight):
            return table[self.rows]
        hit = getattr(self.weight, "_hipad_pair_rows", None)
        stamp = (self.weight.data_ptr(), self.weight._version, tuple(self.weight.shape))
        if hit is None or hit[0] != stamp:
            hit = (stamp, {})
            self.weight._hipad_pair_rows = hit
        pair = hit[1].get(self.rows)
        if pair is None:
            w = self.weight.detach().float()[self.rows[0]:self.rows[1]]
            pair = hit[1][self.rows] = (pack_fragments(w), pack_fragments(w.t()))
        return pair


class ChainSpec:
    """Layers of one chain: [(weight, bias, relu, LayerNorm module or None)], optional trailing Scale parameter."""

    def __init__(self, layers, scale=None):
        self.layers, self.scale = layers, scale
        self.K0, self.N_out = layers[0].K, layers[-1].N

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
                    or m.normalized_shape[0] != layers[-1].N or m.weight is None):
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
            n, k = L.N, L.K
            if n > MAX_WIDTH or k > MAX_WIDTH or L.weight.dtype != torch.float32:
                ok = False
        if scale is not None and scale.numel() != layers[-1].N:
            ok = False
    spec = ChainSpec(layers, scale) if ok else None
    try:
        seq._hipad_chain_spec = spec
    except Exception:  # noqa: BLE001
        pass
    return spec


class OutSlot:
    """Placement of a chain's output inside a tensor shared by several calls of one group: the chain writes rows
    [row0, row0 + M) and columns [col0, col0 + N) of the tensor ``key`` (any hashable) of full shape ``shape``."""

    __slots__ = ("key", "shape", "row0", "col0")

    def __init__(self, key, shape, row0=0, col0=0):
        self.key, self.shape, self.row0, self.col0 = key, tuple(shape), row0, col0


class Call:
    """One chain invocation: out = spec(x0 (+ x1)) * scale (+ residual).

    x0_cols = (lo, hi): the chain reads columns [lo, hi) of ``x0`` (a parent tensor such as the 11-column box anchors); the
                        input gradients of all calls of the group that read the same parent land in ONE gradient tensor.
    scale:              per-column output factor replacing the stack's own trailing Scale parameter (any tensor, e.g.
                        the parameter times a per-frame constant); its gradient is returned to autograd.
    out_slot:           OutSlot when several calls fill parts of one tensor (concatenated encoders, stacked heads)."""

    __slots__ = ("spec", "x0", "x1", "residual", "out_slot", "x0_cols", "scale")

    def __init__(self, spec, x0, x1=None, residual=None, out_slot=None, x0_cols=None, scale=None):
        self.spec, self.x0, self.x1, self.residual = spec, x0, x1, residual
        self.out_slot, self.x0_cols, self.scale = out_slot, x0_cols, scale


def _rows_view(t, width, cols=None):
    """(tensor usable as M rows with a uniform row stride, M, ld): rows of ``width`` floats, or -- with ``cols`` =
    (lo, hi) -- rows of a wider parent of which the caller reads columns lo..hi."""
    if t.dtype != torch.float32:
        t = t.float()
    full = t.shape[-1]
    if cols is None and full != width:
        raise _lib.HipadError(f"chain: input width {full} != layer width {width}")
    ok = t.dim() >= 1 and t.stride(-1) == 1
    ld = full
    if ok and t.dim() >= 2:
        ld = t.stride(-2) if t.shape[-2] > 1 else max(full, 1)
        for d in range(t.dim() - 3, -1, -1):
            if t.shape[d] != 1 and t.stride(d) != t.stride(d + 1) * t.shape[d + 1]:
                ok = False
        if t.dim() >= 3 and t.shape[-2] == 1:
            ok = False  # degenerate row dim: let contiguous() sort it out
        if ld < full:
            ok = False
    if not ok:
        t = t.contiguous()
        ld = full
    M = t.numel() // full
    return t, M, ld


def _acc_target(p, rets, i):
    """In-place accumulation target for parameter ``p`` (its .grad when that is a contiguous fp32 buffer) or a fresh
    zero tensor that is returned to autograd instead."""
    from . import functional as HF
    g = p.grad if (HF.LINEAR_INPLACE_GRAD and p.is_leaf) else None
    if g is not None and g.is_contiguous() and g.dtype == torch.float32:
        HF.INPLACE_PARAMS.add(id(p))
        return g
    rets[i] = torch.zeros_like(p)
    return rets[i]


def _prod(shape):
    n = 1
    for v in shape:
        n *= int(v)
    return n


class _Chains(Function):
    """forward(meta, *tensors): per call the tensors x0, x1, residual, (weight, bias, gamma, beta) per layer, scale."""

    @staticmethod
    def forward(ctx, meta, *tensors):
        calls, slot_keys = meta     # calls: [(spec, out_slot, x0_cols)], slot_keys: ordered keys of the shared outputs
        lib = _lib.load()
        dev = tensors[0].device
        need_grad = any(ctx.needs_input_grad[1:])
        arr = (CChain * len(calls))()
        shared = {}        # key -> tensor (rows_total, width)
        private = []       # outputs of calls without a slot, in call order
        keep, rec = [], []
        pos = 0
        for ci, (spec, slot, x0_cols) in enumerate(calls):
            x0, x1, res = tensors[pos:pos + 3]
            nl = len(spec.layers)
            nparam = 4 * nl + 1
            params = tensors[pos + 3:pos + 3 + nparam]
            pin = pos
            pos += 3 + nparam
            K0, N_out = spec.K0, spec.N_out
            x0v, M, ld0 = _rows_view(x0, K0, x0_cols)
            lo = 0 if x0_cols is None else x0_cols[0]
            if x0_cols is not None and x0_cols[1] - x0_cols[0] != K0:
                raise _lib.HipadError("chain: x0_cols width differs from the first layer's input width")
            x1v = ld1 = None
            if x1 is not None:
                x1v, M1, ld1 = _rows_view(x1, K0)
                if M1 != M:
                    raise _lib.HipadError("chain: x0 and x1 differ in rows")
            if slot is None:
                out = torch.empty(M, N_out, dtype=torch.float32, device=dev)
                private.append(out.view(*x0.shape[:-1], N_out))
                out_ptr, ldo, row0, col0 = out.data_ptr(), N_out, 0, 0
            else:
                width = slot.shape[-1]
                if slot.key not in shared:
                    shared[slot.key] = torch.empty(_prod(slot.shape[:-1]), width, dtype=torch.float32, device=dev)
                out = shared[slot.key]
                row0, col0, ldo = slot.row0, slot.col0, width
                if row0 + M > out.shape[0] or col0 + N_out > width:
                    raise _lib.HipadError("chain: output slot out of range")
                out_ptr = out.data_ptr() + 4 * (row0 * width + col0)
            resv = ldr = None
            if res is not None:
                resv, Mr, ldr = _rows_view(res, N_out)
                if Mr != M:
                    raise _lib.HipadError("chain: residual differs in rows")
            scale = params[-1]
            offs, total = [], 0
            for li, L in enumerate(spec.layers):
                N = L.N
                last = li + 1 == nl
                need_h = need_grad and ((not last) or L.relu or L.ln is not None or scale is not None)
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
            c.x0 = x0v.data_ptr() + 4 * lo
            c.x1 = x1v.data_ptr() if x1v is not None else None
            c.xsum = xsum.data_ptr() if xsum is not None else None
            c.out = out_ptr
            c.out_scale = scale.data_ptr() if scale is not None else None
            if scale is not None and (scale.dtype != torch.float32 or not scale.is_contiguous() or scale.numel() != N_out):
                raise _lib.HipadError("chain: scale must be a contiguous fp32 vector of the output width")
            c.residual = resv.data_ptr() if resv is not None else None
            c.save = save.data_ptr() if save is not None else None
            c.ldx0, c.ldx1, c.ldo, c.ldr = ld0, (ld1 or 0), ldo, (ldr or 0)
            c.M, c.nlayers = M, nl
            shadows = []
            for li, L in enumerate(spec.layers):
                w, b, g, bt = params[4 * li:4 * li + 4]
                pair = L.operands()
                shadows.append(pair)
                cl = c.layers[li]
                cl.w = pair[0].data_ptr()
                cl.bias = b.data_ptr() + 4 * L.r0 if b is not None else None
                cl.gamma = g.data_ptr() if g is not None else None
                cl.beta = bt.data_ptr() if bt is not None else None
                cl.off_h, cl.off_y, cl.off_stats = offs[li]
                cl.N, cl.K = L.N, L.K
                cl.flags = (1 if L.relu else 0) | (2 if L.ln is not None else 0)
                cl.eps = float(L.ln.eps) if L.ln is not None else 0.0
            keep.append((x0v, x1v, resv, shadows, scale))
            rec.append(dict(spec=spec, M=M, ld0=ld0, lo=lo, x0v=x0v, xsum=xsum, save=save, offs=offs, slot=slot, pin=pin,
                            x0_shape=tuple(x0.shape), x0_cols=x0_cols, x0_id=id(x0),
                            x1_shape=None if x1 is None else tuple(x1.shape),
                            res_shape=None if res is None else tuple(res.shape), shadows=shadows, scale=scale))
        with torch.cuda.device(dev):
            _lib.check(lib.hipad_chain_forward(arr, len(calls), _lib.stream_ptr(dev)), "hipad_chain_forward")
        ctx.rec, ctx.n_in, ctx.keep, ctx.calls, ctx.slot_keys = rec, len(tensors), keep, calls, slot_keys
        results = list(private)
        for key in slot_keys:
            shape = next(slot.shape for (_, slot, _) in calls if slot is not None and slot.key == key)
            results.append(shared[key].view(shape))
        return tuple(results)

    @staticmethod
    @once_differentiable
    def backward(ctx, *gouts):
        lib = _lib.load()
        calls, rec, slot_keys = ctx.calls, ctx.rec, ctx.slot_keys
        dev = rec[0]["x0v"].device
        grads = [None] * (ctx.n_in + 1)      # +1: meta
        n_private = sum(1 for (_, slot, _) in calls if slot is None)
        gshared = {}
        for k, key in enumerate(slot_keys):
            g = gouts[n_private + k]
            shape = next(slot.shape for (_, slot, _) in calls if slot is not None and slot.key == key)
            rows, width = _prod(shape[:-1]), shape[-1]
            if g is None:
                g = torch.zeros(rows, width, dtype=torch.float32, device=dev)
            g = g.reshape(rows, width)
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = g.float().contiguous()
            gshared[key] = g
        garr = (CChainGrad * len(calls))()
        dws, keep, parents = [], [], {}
        pi = 0
        for ci, ((spec, slot, x0_cols), r) in enumerate(zip(calls, rec)):
            M, nl, N_out = r["M"], len(spec.layers), spec.N_out
            if slot is None:
                g = gouts[pi]
                pi += 1
                if g is None:
                    g = torch.zeros(M, N_out, dtype=torch.float32, device=dev)
                g2 = g.reshape(M, N_out)
                if g2.dtype != torch.float32 or g2.stride(-1) != 1 or (M > 1 and g2.stride(0) < N_out):
                    g2 = g2.float().contiguous()
                dout_ptr, ldo = g2.data_ptr(), (g2.stride(0) if M > 1 else N_out)
                gres = g2
            else:
                g2 = gshared[slot.key]
                width = g2.shape[1]
                dout_ptr, ldo = g2.data_ptr() + 4 * (slot.row0 * width + slot.col0), width
                gres = g2[slot.row0:slot.row0 + M, slot.col0:slot.col0 + N_out]
            keep.append(g2)
            pin = r["pin"]
            need_x = ctx.needs_input_grad[1 + pin] or (r["x1_shape"] is not None and ctx.needs_input_grad[2 + pin])
            dy_total = sum(M * L.N for L in spec.layers)
            dy = torch.empty(dy_total, dtype=torch.float32, device=dev)
            dx = None
            c = garr[ci]
            if need_x and x0_cols is None:
                dx = torch.empty(M, spec.K0, dtype=torch.float32, device=dev)
                c.dx, c.lddx = dx.data_ptr(), spec.K0
            elif need_x:
                # calls reading column ranges of one parent share its gradient tensor
                ent = parents.get(r["x0_id"])
                if ent is None:
                    D = r["x0_shape"][-1]
                    covered = sorted(rr["x0_cols"] for rr in rec if rr["x0_id"] == r["x0_id"] and rr["x0_cols"] is not None)
                    full, end = True, 0
                    for lo_, hi_ in covered:
                        full, end = full and lo_ <= end, max(end, hi_)
                    full = full and end >= D
                    buf = (torch.empty if full else torch.zeros)(M, D, dtype=torch.float32, device=dev)
                    ent = parents[r["x0_id"]] = buf
                    grads[1 + pin] = buf.view(r["x0_shape"])
                c.dx, c.lddx = ent.data_ptr() + 4 * r["lo"], ent.shape[1]
            else:
                c.dx, c.lddx = None, spec.K0
            rets = [None] * (4 * nl + 1)
            scale = r["scale"]
            c.dout = dout_ptr
            c.out_scale = scale.data_ptr() if scale is not None else None
            c.dscale = None
            if scale is not None and ctx.needs_input_grad[1 + pin + 3 + 4 * nl]:
                c.dscale = _acc_target(scale, rets, 4 * nl).data_ptr()
            c.save = r["save"].data_ptr()
            c.dy = dy.data_ptr()
            c.ldo, c.M, c.nlayers = ldo, M, nl
            off = 0
            for li, L in enumerate(spec.layers):
                N, K = L.N, L.K
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
                        if r["xsum"] is not None:
                            d.x, d.ldx = r["xsum"].data_ptr(), spec.K0
                        else:
                            d.x, d.ldx = r["x0v"].data_ptr() + 4 * r["lo"], r["ld0"]
                    else:
                        poh, poy, _ = r["offs"][li - 1]
                        d.x = r["save"].data_ptr() + 4 * (poy if spec.layers[li - 1].ln is not None else poh)
                        d.ldx = K
                    d.dw = _acc_target(L.weight, rets, 4 * li).data_ptr() + 4 * L.r0 * K
                    d.db = None
                    if L.bias is not None and ctx.needs_input_grad[1 + pin + 3 + 4 * li + 1]:
                        d.db = _acc_target(L.bias, rets, 4 * li + 1).data_ptr() + 4 * L.r0
                    d.M, d.N, d.K = M, N, K
                    dws.append(d)
                off += M * N
            keep.append((dy, dx, rets))
            if dx is not None:
                if ctx.needs_input_grad[1 + pin]:
                    grads[1 + pin] = dx.view(r["x0_shape"])
                if r["x1_shape"] is not None and ctx.needs_input_grad[2 + pin]:
                    grads[2 + pin] = dx.view(r["x1_shape"])
            if r["res_shape"] is not None and ctx.needs_input_grad[3 + pin]:
                grads[3 + pin] = gres.reshape(r["res_shape"])
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

    ``calls``: list of ``Call``.  Returns the outputs of the calls without an ``out_slot`` in call order, followed by the
    shared output tensors in order of first use.  Every input must be a CUDA fp32 tensor."""
    flat, meta, slot_keys = [], [], []
    for c in calls:
        if c.x0_cols is not None and c.x1 is not None:
            raise _lib.HipadError("chain: x0_cols and x1 cannot be combined")
        params = c.spec.params()
        if c.scale is not None:
            params = params[:-1] + [c.scale]
        flat += [c.x0, c.x1, c.residual] + params
        if c.out_slot is not None and c.out_slot.key not in slot_keys:
            slot_keys.append(c.out_slot.key)
        meta.append((c.spec, c.out_slot, c.x0_cols))
    return _Chains.apply((meta, slot_keys), *flat)


def usable(x):
    from . import functional as HF
    return x.is_cuda and HF.LINEAR_MODE == "mfma_bf16" and HF.USE_CHAINS
