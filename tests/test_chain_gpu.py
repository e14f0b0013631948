"""MLP-chain kernels (hip-ad_amd/csrc/chain.hip + chain_dw_kernel in gemm.hip) against (a) the per-layer HIP kernels
they replace (same bf16-operand numerics: tight bound) and (b) plain torch fp32 (BASELINE.json's 1e-2 bf16 class;
gradients compared in the Frobenius norm because a ReLU gate that flips on a pre-activation of 1e-3 moves single
elements by whole dy*w terms, see tests/test_linear_gpu.py).  Shapes are the decoder's: refinement heads on 900 / 100
/ 48 / 5400 rows, box-encoder parts on strided 2..3-column slices written into one concatenated tensor, ragged widths
(11, 10, 2, 1, 40, 24 outputs), one row."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def fro(a, b):
    return float((a.detach().double() - b.detach().double()).norm() / b.detach().double().norm().clamp_min(1e-12))


def build(kind, width=256, in_dims=None, out_dim=None):
    from hipad_amd.compat import Linear, MLPStack, Scale
    from projects.mmdet3d_plugin.models.blocks import linear_relu_ln, mlp_head
    torch.manual_seed(0)
    if kind == "reg":      # refinement regression head (det/blocks.py:93-99): 5 Linear, 2 LayerNorm, Scale
        m = MLPStack(*linear_relu_ln(width, 2, 2), Linear(width, out_dim), Scale([1.0] * out_dim))
        with torch.no_grad():
            m[-1].scale.copy_(torch.linspace(0.5, 1.5, out_dim))
    elif kind == "cls":    # score head: [Linear, ReLU, LayerNorm] x 2, Linear
        m = MLPStack(*linear_relu_ln(width, 1, 2), Linear(width, out_dim))
    elif kind == "enc":    # anchor-encoder part: [Linear, ReLU, LayerNorm] x 2 from a few input columns
        m = MLPStack(*linear_relu_ln(width, 1, 2, in_dims))
    elif kind == "mlp":    # [Linear, ReLU] x 2, Linear
        m = mlp_head(width, out_dim)
    else:
        raise KeyError(kind)
    for name, p in m.named_parameters():   # LayerNorm / bias parameters away from their trivial init
        if p.dim() == 1 and not name.endswith("scale"):
            with torch.no_grad():
                p.add_(torch.randn_like(p) * 0.1)
    return m.cuda()


def grads_of(mod, params_like=None):
    return [p.grad.clone() for p in mod.parameters()]


def run_three_ways(mod, inputs, gout_seed=1, x1=None, residual=None):
    """Outputs and gradients of the chain kernel, the per-layer HIP kernels and torch fp32 for one module call."""
    from hipad_amd import functional as HF
    res = {}
    for mode in ("chain", "layers", "torch"):
        m = copy.deepcopy(mod)
        for p in m.parameters():
            p.grad = torch.full_like(p, 0.25)   # gradients accumulate on top of what is there
        xs = [None if t is None else t.detach().clone().requires_grad_(True) for t in (inputs, x1, residual)]
        HF.USE_CHAINS = mode == "chain"
        try:
            if mode == "torch":
                with HF.linear_mode("torch_fp32"):
                    y = m(xs[0], xs[1], xs[2])
            else:
                y = m(xs[0], xs[1], xs[2])
        finally:
            HF.USE_CHAINS = True
        g = torch.Generator().manual_seed(gout_seed)
        go = torch.randn(y.shape, generator=g).cuda()
        y.backward(go)
        res[mode] = (y.detach(), [None if t is None else t.grad for t in xs], [p.grad - 0.25 for p in m.parameters()])
    return res


def check(res, tight=1e-2, loose=2e-2):
    """The chain forward multiplies hi + lo bf16 halves of the activations (only the weights carry bf16 rounding), so it
    is CLOSER to fp32 than the per-layer kernels, which round both operands: chain vs torch fp32 must be within `loose`
    and no further than the per-layer kernels are; chain vs per-layer differ by the activations' rounding (`tight`).
    Gradients: a ReLU gate that flips between two roundings of a pre-activation moves single elements by whole dy*w
    terms, so they are held in the Frobenius norm to 'no further from torch than the per-layer kernels (x1.5 + 1e-2)'."""
    yc, xc, pc = res["chain"]
    yl, xl, pl = res["layers"]
    yt, xt, pt = res["torch"]
    assert fro(yc, yl) < tight, ("layers", "out", fro(yc, yl))
    assert fro(yc, yt) < loose, ("torch", "out", fro(yc, yt))
    assert fro(yc, yt) < 1.05 * fro(yl, yt) + 1e-4, ("chain further from fp32 than the per-layer kernels", fro(yc, yt), fro(yl, yt))
    for a, b, t in zip(xc, xl, xt):
        if b is not None:
            assert fro(a, t) < max(1.5 * fro(b, t) + 1e-2, 6e-2), ("torch", "dx", fro(a, t), fro(b, t))
    for i, (a, b, t) in enumerate(zip(pc, pl, pt)):
        if float(b.abs().max()) > 0:  # (with a handful of rows one flipped gate is a visible share of a weight gradient)
            assert fro(a, t) < max(1.5 * fro(b, t) + 1e-2, 6e-2), ("torch", "param", i, tuple(b.shape), fro(a, t), fro(b, t))


@pytest.mark.parametrize("kind,M,out_dim", [("reg", 900, 11), ("reg", 100, 40), ("reg", 48, 12), ("cls", 900, 10),
                                             ("cls", 900, 2), ("cls", 144, 1), ("mlp", 5400, 24), ("mlp", 1, 6),
                                             ("cls", 5400, 1), ("reg", 17, 11)])
def test_head_chains(kind, M, out_dim):
    mod = build(kind, 256, out_dim=out_dim)
    g = torch.Generator().manual_seed(M + out_dim)
    x = torch.randn(1, M, 256, generator=g).cuda()
    check(run_three_ways(mod, x))


def test_input_sum_and_residual_are_fused():
    mod = build("reg", 256, out_dim=12)
    g = torch.Generator().manual_seed(3)
    x, x1, res = (torch.randn(2, 240, n, generator=g).cuda() for n in (256, 256, 12))
    out = run_three_ways(mod, x, x1=x1, residual=res)
    check(out)
    # both summands receive the same gradient; the residual receives the output gradient
    assert torch.equal(out["chain"][1][0], out["chain"][1][1])
    assert fro(out["chain"][1][2], out["torch"][1][2]) < 1e-6


@pytest.mark.parametrize("width,in_dims,lo", [(128, 3, 0), (32, 3, 3), (32, 2, 6), (64, 3, 8), (256, 40, 0), (256, 12, 0)])
def test_encoder_parts_on_strided_column_slices(width, in_dims, lo):
    mod = build("enc", width, in_dims=in_dims)
    g = torch.Generator().manual_seed(width + lo)
    full = torch.randn(1, 900, max(11, in_dims), generator=g).cuda()
    from hipad_amd import functional as HF
    res = {}
    for mode in ("chain", "layers", "torch"):
        m = copy.deepcopy(mod)
        for p in m.parameters():
            p.grad = torch.zeros_like(p)
        a = full.clone().requires_grad_(True)
        HF.USE_CHAINS = mode == "chain"
        try:
            if mode == "torch":
                with HF.linear_mode("torch_fp32"):
                    y = m(a[..., lo:lo + in_dims])
            else:
                y = m(a[..., lo:lo + in_dims])
        finally:
            HF.USE_CHAINS = True
        y.backward(torch.ones_like(y) * torch.linspace(-1, 1, width).cuda())
        res[mode] = (y.detach(), [a.grad], [p.grad for p in m.parameters()])
    check(res, tight=1e-2, loose=3e-2)


def test_group_of_chains_writes_one_concatenated_tensor():
    """The four parts of the box encoder (det/blocks.py:22-74, mode "cat") as ONE launch into one (M, 256) tensor."""
    from hipad_amd import chain as CH
    widths, cols = [128, 32, 32, 64], [(0, 3), (3, 6), (6, 8), (8, 11)]
    mods = [build("enc", w, in_dims=b - a) for w, (a, b) in zip(widths, cols)]
    g = torch.Generator().manual_seed(11)
    anchor = torch.randn(2, 450, 11, generator=g).cuda()
    go = torch.randn(2, 450, 256, generator=g).cuda()
    # chain group
    a1 = anchor.clone().requires_grad_(True)
    for m in mods:
        for p in m.parameters():
            p.grad = torch.zeros_like(p)
    col = 0
    calls = []
    for m, w, (lo, hi) in zip(mods, widths, cols):
        calls.append(CH.Call(CH.spec_of(m), a1, x0_cols=(lo, hi), out_slot=CH.OutSlot("cat", (2, 450, 256), col0=col)))
        col += w
    (out,) = CH.run(calls)
    assert out.shape == (2, 450, 256)
    out.backward(go)
    got = (out.detach(), a1.grad.clone(), [p.grad.clone() for m in mods for p in m.parameters()])
    # per-layer kernels + torch.cat
    from hipad_amd import functional as HF
    ref_mods = [copy.deepcopy(m) for m in mods]
    for m in ref_mods:
        for p in m.parameters():
            p.grad = torch.zeros_like(p)
    a2 = anchor.clone().requires_grad_(True)
    HF.USE_CHAINS = False
    try:
        ref = torch.cat([m(a2[..., lo:hi]) for m, (lo, hi) in zip(ref_mods, cols)], dim=-1)
    finally:
        HF.USE_CHAINS = True
    ref.backward(go)
    assert fro(got[0], ref) < 1e-2
    assert fro(got[1], a2.grad) < 8e-2          # gate flips between the two roundings, see check()
    for a, b in zip(got[2], [p.grad for m in ref_mods for p in m.parameters()]):
        assert fro(a, b) < 8e-2, tuple(b.shape)


def test_more_than_eight_chains_and_no_grad():
    from hipad_amd import chain as CH
    mods = [build("cls", 256, out_dim=1) for _ in range(11)]
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(1, 48, 256, generator=g).cuda() for _ in mods]
    with torch.no_grad():
        outs = CH.run([CH.Call(CH.spec_of(m), x) for m, x in zip(mods, xs)])
        from hipad_amd import functional as HF
        HF.USE_CHAINS = False
        try:
            refs = [m(x) for m, x in zip(mods, xs)]
        finally:
            HF.USE_CHAINS = True
    assert len(outs) == 11
    for a, b in zip(outs, refs):
        assert fro(a, b) < 1e-2


def test_unsupported_stacks_fall_back_to_layers():
    from hipad_amd import chain as CH
    from hipad_amd.compat import Linear, MLPStack
    wide = MLPStack(Linear(256, 512), Linear(512, 256)).cuda()
    assert CH.spec_of(wide) is None
    y = wide(torch.randn(4, 256).cuda())
    assert y.shape == (4, 256)


def test_row_slots_scale_override_and_shared_parent_gradient():
    """Two chains stack their rows into one tensor (OutSlot.row0); a call-level scale replaces the stack's Scale and
    receives a gradient; chains reading column ranges of one parent return ONE gradient tensor for it."""
    from hipad_amd import chain as CH
    from hipad_amd import functional as HF
    g = torch.Generator().manual_seed(21)
    m1, m2 = build("reg", 256, out_dim=12), build("reg", 256, out_dim=12)
    for m in (m1, m2):
        for p in m.parameters():
            p.grad = torch.zeros_like(p)
    x1 = torch.randn(1, 48, 256, generator=g).cuda().requires_grad_(True)
    x2 = torch.randn(1, 144, 256, generator=g).cuda().requires_grad_(True)
    col = torch.linspace(0.5, 2.0, 12).cuda()
    s2 = (m2[-1].scale * col)
    go = torch.randn(1, 192, 12, generator=g).cuda()
    (out,) = CH.run([CH.Call(CH.spec_of(m1), x1, out_slot=CH.OutSlot("o", (1, 192, 12), row0=0)),
                     CH.Call(CH.spec_of(m2), x2, out_slot=CH.OutSlot("o", (1, 192, 12), row0=48), scale=s2)])
    out.backward(go)
    got = (out.detach(), x1.grad.clone(), x2.grad.clone(), m2[-1].scale.grad.clone(), m1[-1].scale.grad.clone())
    r1, r2 = copy.deepcopy(m1), copy.deepcopy(m2)
    for m in (r1, r2):
        for p in m.parameters():
            p.grad = torch.zeros_like(p)
    y1, y2 = x1.detach().clone().requires_grad_(True), x2.detach().clone().requires_grad_(True)
    with HF.linear_mode("torch_fp32"):
        a = r1(y1)
        b = r2[:-1](y2) * (r2[-1].scale * col)
    ref = torch.cat([a, b], dim=1)
    ref.backward(go)
    assert fro(got[0], ref) < 2e-2
    errs = [fro(got[1], y1.grad), fro(got[2], y2.grad), fro(got[3], r2[-1].scale.grad), fro(got[4], r1[-1].scale.grad)]
    assert all(e < 0.12 for e in errs), errs   # five bf16 layers with ReLU gates: see check()


def test_chunk_mix_and_motion_embedding_match_torch():
    from hipad_amd import functional as HF
    from hipad_amd import lib
    from projects.mmdet3d_plugin.models.attention import gen_sineembed_for_position
    g = torch.Generator().manual_seed(8)
    x0 = torch.randn(2, 480, 256, generator=g).cuda().requires_grad_(True)
    x1 = torch.randn(2, 480, 256, generator=g).cuda().requires_grad_(True)
    table = [[1, 1, 1, 1, 0, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, 0, 0, 1, 0, 0], [1, 1, 1, 1, 0, 1, 0, 0, 1, 0],
             [1, 1, 1, 1, 0, 0, 1, 0, 0, 1]]
    out = HF.chunk_mix(x0, x1, table, 48)
    go = torch.randn(out.shape, generator=g).cuda()
    out.backward(go)
    y0, y1 = x0.detach().cpu().requires_grad_(True), x1.detach().cpu().requires_grad_(True)
    ref = HF.chunk_mix(y0, y1, table, 48)       # the torch expression (CPU branch)
    ref.backward(go.cpu())
    assert float((out.cpu() - ref).abs().max()) < 1e-5
    assert float((x0.grad.cpu() - y0.grad).abs().max()) < 1e-5 and torch.equal(x0.grad, x1.grad)
    # motion-mode embedding vs the torch expression of the reference
    cls = torch.randn(1, 900, 9, generator=g).cuda()
    box = torch.randn(1, 900, 11, generator=g).cuda()
    anchors = (torch.randn(9, 6, 6, 2, generator=g) * 20).cuda()
    idx = torch.arange(128, dtype=torch.float32).cuda()
    freq = 10000 ** (2 * torch.div(idx, 2, rounding_mode="floor") / 128)
    got = lib.motion_query_embed(cls, box, anchors, freq, 6, 7)
    modes = anchors[cls.argmax(dim=-1)]
    yaw = torch.atan2(box[..., 6], box[..., 7])
    c, s = yaw.cos()[..., None, None], yaw.sin()[..., None, None]
    x, y = modes[..., 0], modes[..., 1]
    pts = torch.stack([c * x - s * y, s * x + c * y], dim=-1)
    want = gen_sineembed_for_position(pts[..., -1, :])
    assert got.shape == want.shape
    # sin / cos of arguments up to ~1e3 rad: a last-bit difference in the argument moves the value by ~1e-4
    assert float((got - want).abs().max()) < 2e-3 and float((got - want).abs().mean()) < 1e-5


def test_step_offsets_match_the_slice_and_concatenate_expression():
    """Way-points -> offsets (plan branch): bit-equal to the torch expression, forward and backward."""
    from hipad_amd import functional as HF
    from hipad_amd import lib
    g = torch.Generator().manual_seed(9)
    for shape in [(2, 1, 48, 6, 2), (1, 5, 1, 3), (3, 7, 1)]:
        x = torch.randn(*shape, generator=g).cuda().requires_grad_(True)
        out = HF.step_offsets(x)
        go = torch.randn(shape, generator=g).cuda()
        out.backward(go)
        y = x.detach().cpu().requires_grad_(True)
        ref = HF.step_offsets(y)                  # CPU branch: cat(x[:1], x[1:] - x[:-1])
        ref.backward(go.cpu())
        assert torch.equal(out.cpu(), ref) and torch.equal(x.grad.cpu(), y.grad)
    with pytest.raises(lib.HipadError):
        lib.step_offsets(torch.zeros(4, device="cuda"))
    with pytest.raises(lib.HipadError):
        lib.step_offsets(torch.zeros(2, 3, 2, device="cuda", dtype=torch.float64))


def test_add_rows_matches_broadcast_additions():
    """base + row vectors broadcast over a sample's rows: forward equal to the chained torch additions (same order of
    fp32 additions: bit-equal), one shared row-sum gradient for the vectors."""
    from hipad_amd import functional as HF
    g = torch.Generator().manual_seed(10)
    for bs, N, C, k in [(2, 480, 256, 3), (1, 7, 64, 1), (3, 901, 256, 2)]:
        base = torch.randn(bs, N, C, generator=g).cuda().requires_grad_(True)
        rows = [torch.randn(bs, 1, C, generator=g).cuda().requires_grad_(True) for _ in range(k)]
        out = HF.add_rows(base, *rows)
        go = torch.randn(out.shape, generator=g).cuda()
        out.backward(go)
        b2 = base.detach().clone().requires_grad_(True)
        r2 = [r.detach().clone().requires_grad_(True) for r in rows]
        ref = b2
        for r in r2:
            ref = ref + r
        ref.backward(go)
        assert torch.equal(out, ref) and torch.equal(base.grad, b2.grad)
        for a, b in zip(rows, r2):
            assert a.grad.shape == b.grad.shape
            assert float((a.grad - b.grad).abs().max()) <= 1e-5 * float(b.grad.abs().max())
    # None entries are skipped; shapes the kernel does not take fall back to torch additions
    x = torch.randn(2, 5, 6, generator=g).cuda()
    assert torch.equal(HF.add_rows(x, None, None), x)
    r = torch.randn(2, 1, 6, generator=g).cuda()
    assert torch.equal(HF.add_rows(x, r), x + r)


def test_keep_mask_statistics_and_clock():
    from hipad_amd import functional as HF
    from hipad_amd import lib
    clock = HF.dropout_clock("cuda")
    a = lib.keep_mask((1, 900, 6, 13), 0.15, 1234, clock, torch.device("cuda"))
    b = lib.keep_mask((1, 900, 6, 13), 0.15, 1234, clock, torch.device("cuda"))
    assert torch.equal(a, b)                                   # same (seed, clock) -> same mask
    vals = torch.unique(a)
    assert vals.numel() == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1 / 0.85) < 1e-6
    assert abs(float((a == 0).float().mean()) - 0.15) < 0.01   # 70 200 draws: sigma = 0.0013
    HF.advance_dropout_clock("cuda")
    c = lib.keep_mask((1, 900, 6, 13), 0.15, 1234, clock, torch.device("cuda"))
    assert 0.2 < float((a != c).float().mean()) < 0.3         # independent masks differ in 2 p (1 - p) = 25.5 % of the slots
    d = lib.keep_mask((1, 900, 6, 13), 0.15, 99, clock, torch.device("cuda"))
    assert 0.2 < float((c != d).float().mean()) < 0.3


@pytest.mark.parametrize("mode", ["self", "cross"])
def test_attention_projections_on_row_blocks_of_the_packed_weight(mode):
    """FlashMHA's three in-projections as one grouped launch of single-layer chains on the 256-row blocks of
    in_proj_weight, positional inputs added inside the kernel: same values and gradients (inputs, positional inputs,
    the packed weight and bias) as fp64 torch, within the bf16 class; the module's output equals the per-layer path."""
    from hipad_amd import functional as HF
    from projects.mmdet3d_plugin.models.attention import MultiheadFlashAttention
    torch.manual_seed(3)
    E, B, Nq, Nk = 256, 1, 333, 517
    attn = MultiheadFlashAttention(E, 8, attn_drop=0.0, dropout_layer=dict(type="Dropout", drop_prob=0.0)).cuda()
    with torch.no_grad():
        attn.attn.in_proj_bias.normal_(0, 0.1)
    q = torch.randn(B, Nq, E, device="cuda", requires_grad=True)
    qpos = torch.randn(B, Nq, E, device="cuda", requires_grad=True)
    if mode == "self":
        k = kpos = v = None
    else:
        k = torch.randn(B, Nk, E, device="cuda", requires_grad=True)
        kpos = torch.randn(B, Nk, E, device="cuda", requires_grad=True)
        v = torch.randn(B, Nk, E, device="cuda", requires_grad=True)
    leaves = [t for t in (q, qpos, k, kpos, v) if t is not None] + [attn.attn.in_proj_weight, attn.attn.in_proj_bias]

    def project(use_chains):
        kk, kp, vv = (q, qpos, q) if mode == "self" else (k, kpos, v)
        if use_chains:
            return attn.attn._project_chains(q, kk, vv, qpos, kp)
        qi = q + qpos
        ki = qi if mode == "self" else kk + kp
        return attn.attn._project(qi, ki, vv)

    def ref64():
        W, b = attn.attn.in_proj_weight.double(), attn.attn.in_proj_bias.double()
        kk, kp, vv = (q, qpos, q) if mode == "self" else (k, kpos, v)
        ins = ((q + qpos).double(), (kk + kp).double(), vv.double())
        return [x @ W[i * E:(i + 1) * E].t() + b[i * E:(i + 1) * E] for i, x in enumerate(ins)]

    gouts = [torch.randn(B, Nq, E, device="cuda")] + [torch.randn(B, Nq if mode == "self" else Nk, E, device="cuda") for _ in range(2)]

    def grads(outs):
        for t in leaves:
            t.grad = None
        torch.autograd.backward([o.float() for o in outs], gouts)
        return [t.grad.double().clone() for t in leaves]

    want = ref64()
    gwant = grads(want)
    for use_chains in (True, False):
        outs = project(use_chains)
        for o, w in zip(outs, want):
            assert o.is_contiguous() or not use_chains
            assert fro(o.double(), w) < 1e-2
        for g, gw in zip(grads(outs), gwant):
            assert fro(g, gw) < 1e-2
    # the whole module: chains on and off agree within the bf16 class
    outs = {}
    for flag in (True, False):
        HF.USE_CHAINS = flag
        try:
            outs[flag] = attn(query=q, key=k, value=v, query_pos=qpos, key_pos=kpos)
        finally:
            HF.USE_CHAINS = True
    assert fro(outs[True], outs[False]) < 1e-2


def test_dropout_add_is_inverted_dropout_with_a_replayable_mask():
    """identity + dropout(x) in one launch: every element is either identity or identity + x / (1 - p), the keep rate is
    1 - p, the backward applies the SAME mask to the output gradient, the mask changes with the dropout clock and with
    the call-site seed, and p = 0 / eval are the plain sum."""
    from hipad_amd import functional as HF
    torch.manual_seed(0)
    x = torch.randn(2, 700, 256, device="cuda", requires_grad=True)
    idt = torch.randn(2, 700, 256, device="cuda", requires_grad=True)
    p = 0.1
    y = HF.dropout_add(x, idt, p, 12345)
    kept = (y - idt).detach().abs() > 0
    assert torch.allclose((y - idt)[kept], (x / (1 - p))[kept], rtol=1e-6, atol=1e-6)
    rate = float(kept.float().mean())
    assert abs(rate - (1 - p)) < 5e-3, rate
    g = torch.randn_like(y)
    y.backward(g)
    assert torch.equal(idt.grad, g)
    assert torch.allclose(x.grad, torch.where(kept, g / (1 - p), torch.zeros_like(g)), rtol=1e-6, atol=0)
    y2 = HF.dropout_add(x, idt, p, 12345)
    assert torch.equal(y2, y)                                   # same clock, same seed: same mask (graph replay of a backward)
    HF.advance_dropout_clock(x.device)
    y3 = HF.dropout_add(x, idt, p, 12345)
    y4 = HF.dropout_add(x, idt, p, 54321)
    for other in (y3, y4):
        changed = float((((other - idt).abs() > 0) != kept).float().mean())
        assert 0.1 < changed < 0.26, changed                    # independent masks differ on 2 p (1 - p) = 18 % of elements
    assert torch.equal(HF.dropout_add(x, idt, 0.0, 1), idt + x) and torch.equal(HF.dropout_add(x, idt, p, 1, training=False), idt + x)
