"""GPU parity tests of the deformable aggregation kernels (through the C ABI) against
  (1) the golden vectors made by the reference's own PyTorch fallback (tests/golden),
  (2) the CPU oracle (oracle/daf_oracle.c) on seeded inputs it finishes in seconds,
  (3) size-independent properties at the full BASELINE sizes.
Tolerances: index work bit-exact; values 1e-3 relative fp32 (north star) -- asserted tighter
where the arithmetic allows."""
import numpy as np
import pytest
import torch

from oracle import daf as O

pytestmark = pytest.mark.gpu
CASES = ["daf_unit", "daf_multicam", "daf_ragged"]
REL = 1e-3  # BASELINE.json north_star: within 1e-3 rel fp32


def dev(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    return t.to(dtype) if dtype is not None else t


def rel_err(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def load_case(golden, case):
    z = golden(case)
    t = dict(feat=dev(z["feat"]), ss=dev(z["spatial_shape"], torch.int32), st=dev(z["scale_start_index"], torch.int32),
             loc=dev(z["loc"]), w=dev(z["weights"]), gout=dev(z["grad_out"]))
    return z, t


def off_kink(loc, spatial_shape, eps=1e-4):
    ok = np.ones(loc.shape[:-1], bool)
    for c in range(loc.shape[3]):
        for h, w in spatial_shape[c]:
            for k, size in ((0, w), (1, h)):
                pix = loc[:, :, :, c, k].astype(np.float64) * size - 0.5
                ok[:, :, :, c] &= np.abs(pix - np.round(pix)) > eps
    return ok


@pytest.fixture(scope="module")
def lib():
    from hipad_amd import lib as L
    L.load()
    return L


@pytest.mark.parametrize("case", CASES)
def test_forward_vs_golden(golden, lib, case):
    z, t = load_case(golden, case)
    out = lib.daf_forward(t["feat"], t["ss"], t["st"], t["loc"], t["w"])
    assert rel_err(out, z["out"]) < 2e-5


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("overwrite", [False, True])
@pytest.mark.parametrize("atomic_feat", [False, True])
def test_backward_vs_golden(golden, lib, case, overwrite, atomic_feat):
    z, t = load_case(golden, case)
    gf = torch.zeros_like(t["feat"])
    if overwrite:  # poison: the kernel must write every element itself
        gl = torch.full_like(t["loc"], float("nan"))
        gw = torch.full_like(t["w"], float("nan"))
    else:
        gl, gw = torch.zeros_like(t["loc"]), torch.zeros_like(t["w"])
    lib.daf_backward(t["feat"], t["ss"], t["st"], t["loc"], t["w"], t["gout"], gf, gl, gw, overwrite_loc_w=overwrite,
                     atomic_feat=atomic_feat)
    assert rel_err(gf, z["grad_feat"]) < 2e-5
    assert rel_err(gw, z["grad_weights"]) < 2e-5
    ok = off_kink(z["loc"], z["spatial_shape"])
    assert rel_err(gl.cpu().numpy()[ok], z["grad_loc"][ok]) < 5e-5
    loc = z["loc"]
    dropped = ~((loc[..., 0] > 0) & (loc[..., 0] < 1) & (loc[..., 1] > 0) & (loc[..., 1] < 1))
    assert np.all(gl.cpu().numpy()[dropped] == 0) and np.all(gw.cpu().numpy()[dropped] == 0)


def test_backward_accumulates_in_reference_mode(golden, lib):
    """flags=0: all three gradients are added to what the caller passed (cpp:86-124 contract)."""
    z, t = load_case(golden, "daf_multicam")
    gf, gl, gw = torch.ones_like(t["feat"]), torch.ones_like(t["loc"]), torch.ones_like(t["w"])
    lib.daf_backward(t["feat"], t["ss"], t["st"], t["loc"], t["w"], t["gout"], gf, gl, gw)
    assert rel_err(gf - 1, z["grad_feat"]) < 1e-4
    assert rel_err(gw - 1, z["grad_weights"]) < 1e-4


def make_inputs(seed, bs, A, P, shapes, cams=6, C=256, G=8, lo=-0.2, hi=1.2):
    g = torch.Generator().manual_seed(seed)
    ss = np.array([shapes] * cams, np.int32)
    sizes = (ss[..., 0] * ss[..., 1]).reshape(-1)
    st = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32).reshape(cams, len(shapes))
    F = int(sizes.sum())
    feat = torch.randn(bs, F, C, generator=g)
    loc = torch.rand(bs, A, P, cams, 2, generator=g) * (hi - lo) + lo
    w = torch.softmax(torch.randn(bs, A, P * cams * len(shapes), G, generator=g), 2).reshape(bs, A, P, cams, len(shapes), G)
    gout = torch.randn(bs, A, C, generator=g)
    return feat, ss, st, loc.contiguous(), w.contiguous(), gout


@pytest.mark.parametrize("A,P", [(37, 13), (5, 300), (9, 90), (1, 13)])
@pytest.mark.parametrize("atomic_feat", [False, True])
def test_vs_cpu_oracle_seeded(lib, A, P, atomic_feat):
    """det / map / plan / ego point counts on a quarter-resolution 6-cam 4-level pyramid."""
    feat, ss, st, loc, w, gout = make_inputs(10 + P, 2, A, P, [(16, 44), (8, 22), (4, 11), (2, 6)])
    ref = O.daf_forward(feat.numpy(), ss, st, loc.numpy(), w.numpy(), acc64=True)
    rgf, rgl, rgw = O.daf_backward(feat.numpy(), ss, st, loc.numpy(), w.numpy(), gout.numpy(), acc64=True)
    d = [x.cuda() for x in (feat, torch.from_numpy(ss), torch.from_numpy(st), loc, w, gout)]
    out = lib.daf_forward(*d[:5])
    assert rel_err(out, ref) < 1e-5
    gf = torch.zeros_like(d[0]); gl = torch.empty_like(d[3]); gw = torch.empty_like(d[4])
    lib.daf_backward(*d, gf, gl, gw, overwrite_loc_w=True, atomic_feat=atomic_feat)
    assert rel_err(gf, rgf) < 1e-5
    assert rel_err(gw, rgw) < 1e-5
    assert rel_err(gl, rgl) < 1e-4  # same floor() as the oracle: no kink exclusion needed


def test_index_work_bit_exact(lib):
    """valid flags, integer corners, bounds masks and row bases equal the oracle's bit for bit."""
    feat, ss, st, loc, w, gout = make_inputs(3, 2, 64, 13, [(64, 176), (32, 88), (16, 44), (8, 22)], lo=-0.05, hi=1.05)
    # plant values that sit exactly on pixel centres / borders
    flat = loc.view(-1, 2)
    for i, (x, y) in enumerate([(0.5 / 176, 0.5 / 64), (1.5 / 176, 2.5 / 64), (175.5 / 176, 63.5 / 64), (0.0, 0.5),
                                (1.0, 0.5), (0.5, 1.0), (np.nextafter(np.float32(1), np.float32(0)), 0.5),
                                (np.nextafter(np.float32(0), np.float32(1)), 0.5), (1e-30, 1e-30)]):
        flat[7 * i] = torch.tensor([float(x), float(y)])
    v_ref, t_ref = O.daf_taps(ss, st, loc.numpy(), feat.shape[1])
    v, t = lib.daf_taps(torch.from_numpy(ss).cuda(), torch.from_numpy(st).cuda(), loc.cuda(), feat.shape[1])
    assert np.array_equal(v.cpu().numpy(), v_ref)
    assert np.array_equal(t.cpu().numpy(), t_ref)


def test_forward_is_deterministic_and_chunking_invariant(lib):
    feat, ss, st, loc, w, gout = make_inputs(5, 1, 100, 300, [(64, 176), (32, 88), (16, 44), (8, 22)])
    d = [x.cuda() for x in (feat, torch.from_numpy(ss), torch.from_numpy(st), loc, w)]
    a = lib.daf_forward(*d)
    b = lib.daf_forward(*d)
    assert torch.equal(a, b)  # no atomics: bitwise reproducible
    L = lib.load()
    try:
        L.hipad_daf_set_pairs_per_wave(24, 24)
        c = lib.daf_forward(*d)
    finally:
        L.hipad_daf_set_pairs_per_wave(0, 0)
    assert rel_err(c, a.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("A,P,name", [(900, 13, "det"), (100, 300, "map"), (480, 90, "plan")])
def test_full_size_properties(lib, A, P, name):
    """BASELINE sizes (704x256 pyramid, stage2 query counts): linearity in the weights and in the
    features, zero weights -> zero, and <grad_w, w> == <out, grad_out> (Euler identity: out is
    linear in w), which ties backward to forward without a CPU reference."""
    feat, ss, st, loc, w, gout = make_inputs(7, 1, A, P, [(64, 176), (32, 88), (16, 44), (8, 22)])
    d = [x.cuda() for x in (feat, torch.from_numpy(ss), torch.from_numpy(st), loc, w, gout)]
    out = lib.daf_forward(*d[:5])
    assert torch.count_nonzero(lib.daf_forward(d[0], d[1], d[2], d[3], torch.zeros_like(d[4]))) == 0
    out2 = lib.daf_forward(d[0], d[1], d[2], d[3], (2.5 * d[4]).contiguous())
    assert rel_err(out2, (2.5 * out).cpu().numpy()) < 1e-5
    out3 = lib.daf_forward((d[0] * -3).contiguous(), *d[1:5])
    assert rel_err(out3, (-3 * out).cpu().numpy()) < 1e-5
    gf = torch.zeros_like(d[0]); gl = torch.empty_like(d[3]); gw = torch.empty_like(d[4])
    lib.daf_backward(*d, gf, gl, gw, overwrite_loc_w=True)
    lhs = float((gw.double() * d[4].double()).sum())
    rhs = float((out.double() * d[5].double()).sum())
    assert abs(lhs - rhs) < 1e-4 * max(1.0, abs(rhs))
    lhs_f = float((gf.double() * d[0].double()).sum())  # out is linear in feat too
    assert abs(lhs_f - rhs) < 1e-4 * max(1.0, abs(rhs))
    # the sorted path and the atomic-scatter kernel are two implementations of the same sum
    gf2 = torch.zeros_like(d[0]); gl2 = torch.empty_like(d[3]); gw2 = torch.empty_like(d[4])
    lib.daf_backward(*d, gf2, gl2, gw2, overwrite_loc_w=True, atomic_feat=True)
    assert rel_err(gf2, gf.cpu().numpy()) < 1e-5
    assert rel_err(gw2, gw.cpu().numpy()) < 1e-5
    assert rel_err(gl2, gl.cpu().numpy()) < 1e-4


def test_grad_loc_finite_difference(lib):
    """grad_loc against central differences of the forward, away from pixel-centre kinks."""
    feat, ss, st, loc, w, gout = make_inputs(11, 1, 6, 5, [(16, 44), (8, 22)], cams=2, lo=0.1, hi=0.9)
    ok = off_kink(loc.numpy(), ss, eps=2e-2)
    d = [x.cuda() for x in (feat, torch.from_numpy(ss), torch.from_numpy(st), loc, w, gout)]
    gl = torch.empty_like(d[3])
    lib.daf_backward(*d, None, gl, None, overwrite_loc_w=True)
    eps = 1e-4
    idx = np.argwhere(ok)[:12]
    for b, a, p, c in idx:
        for k in range(2):
            lp, lm = loc.clone(), loc.clone()
            lp[b, a, p, c, k] += eps
            lm[b, a, p, c, k] -= eps
            fp = lib.daf_forward(d[0], d[1], d[2], lp.cuda(), d[4]).double()
            fm = lib.daf_forward(d[0], d[1], d[2], lm.cuda(), d[4]).double()
            fd = float(((fp - fm) * d[5].double()).sum() / (2 * eps))
            got = float(gl[b, a, p, c, k])
            assert abs(fd - got) < 2e-2 * max(1.0, abs(fd)), (b, a, p, c, k, fd, got)


def test_plugin_op_autograd_matches_golden(golden):
    """The drop-in surface: projects.mmdet3d_plugin.ops.deformable_aggregation_function."""
    from projects.mmdet3d_plugin.ops import deformable_aggregation_function as DAF
    z = golden("daf_multicam")
    feat = dev(z["feat"]).requires_grad_(True)
    loc = dev(z["loc"]).requires_grad_(True)
    w = dev(z["weights"]).requires_grad_(True)
    ss, st = dev(z["spatial_shape"]), dev(z["scale_start_index"])  # int64, as feature_maps_format emits
    out = DAF(feat, ss, st, loc, w)
    out.backward(dev(z["grad_out"]))
    assert rel_err(out, z["out"]) < 2e-5
    assert rel_err(feat.grad, z["grad_feat"]) < 2e-5
    assert rel_err(w.grad, z["grad_weights"]) < 2e-5


def test_shared_feature_grad_sink(golden):
    from projects.mmdet3d_plugin.ops import deformable_aggregation_function as DAF, shared_feature_grad
    z = golden("daf_multicam")
    feat = dev(z["feat"]).requires_grad_(True)
    ss, st = dev(z["spatial_shape"]), dev(z["scale_start_index"])
    loc, w, go = dev(z["loc"]), dev(z["weights"]).requires_grad_(True), dev(z["grad_out"])
    shared = shared_feature_grad(feat * 1.0)
    y = DAF(shared, ss, st, loc, w) + 2.0 * DAF(shared, ss, st, loc, w)
    y.backward(go)
    assert rel_err(feat.grad, 3.0 * z["grad_feat"]) < 5e-5
    assert rel_err(w.grad, 3.0 * z["grad_weights"]) < 5e-5


def test_cpu_tensors_fail_loudly(lib, golden):
    z = golden("daf_unit")
    with pytest.raises(lib.HipadError):
        lib.daf_forward(torch.from_numpy(z["feat"]), dev(z["spatial_shape"], torch.int32),
                        dev(z["scale_start_index"], torch.int32), dev(z["loc"]), dev(z["weights"]))


def test_bad_dims_return_error(lib):
    L = lib.load()
    assert L.hipad_daf_forward(None, None, None, None, None, None, 1, 6, 100, 256, 4, 10, 13, 8, None, 0, None) == -1
    assert L.hipad_daf_forward(None, None, None, None, None, None, 1, 6, 100, 250, 4, 10, 13, 8, None, 0, None) == -1


@pytest.mark.parametrize("kind", ["det", "map", "plan"])
def test_bf16_feature_rows_are_bit_identical_to_the_widened_tensor(lib, kind):
    """hipad_daf_forward_bf16 / _backward_bf16 read the encoder's bf16 rows directly; on a pyramid whose values are bf16
    the results (output, grad_loc, grad_w, grad_feat) equal the fp32 entries on the widened tensor bit for bit -- the
    arithmetic is the same, only the load is half as wide.  Shapes of the stage-2 calls, bs 2 for one of them; through
    the op's autograd surface as well (grad of a bf16 feature tensor comes back as bf16)."""
    from projects.mmdet3d_plugin.ops import deformable_aggregation_function as DAF
    torch.manual_seed(7)
    shapes = [(64, 176), (32, 88), (16, 44), (8, 22)]
    cams, C, G = 6, 256, 8
    bs, A, P = {"det": (2, 900, 13), "map": (1, 100, 300), "plan": (1, 480, 90)}[kind]
    ss = torch.tensor([shapes] * cams, dtype=torch.int32, device="cuda")
    sizes = (ss[..., 0] * ss[..., 1]).reshape(-1)
    st = (sizes.cumsum(0) - sizes).reshape(cams, len(shapes)).int()
    F_ = int(sizes.sum())
    feat16 = torch.randn(bs, F_, C, device="cuda").to(torch.bfloat16)
    feat32 = feat16.float()
    loc = torch.rand(bs, A, P, cams, 2, device="cuda") * 1.6 - 0.3
    w = torch.softmax(torch.randn(bs, A, P * cams * len(shapes), G, device="cuda"), 2).reshape(bs, A, P, cams, len(shapes), G).contiguous()
    gout = torch.randn(bs, A, C, device="cuda")
    o32 = lib.daf_forward(feat32, ss, st, loc, w)
    o16 = lib.daf_forward(feat16, ss, st, loc, w)
    assert torch.equal(o32, o16)
    res = {}
    for name, f in (("f32", feat32), ("bf16", feat16)):
        gf = torch.zeros(bs, F_, C, device="cuda")
        gl, gw = torch.full_like(loc, float("nan")), torch.full_like(w, float("nan"))
        lib.daf_backward(f, ss, st, loc, w, gout, gf, gl, gw, overwrite_loc_w=True)
        res[name] = (gf, gl, gw)
    # grad_loc / grad_w: bit for bit.  grad_feat: rows that straddle two tap batches are finished with fp32 atomics, whose
    # order differs from launch to launch (two launches on the SAME input differ as much): equal to rounding
    assert torch.equal(res["f32"][1], res["bf16"][1]) and torch.equal(res["f32"][2], res["bf16"][2])
    assert float((res["f32"][0] - res["bf16"][0]).abs().max() / res["f32"][0].abs().max()) < 1e-6
    # autograd surface
    f = feat16.clone().requires_grad_(True)
    l2, w2 = loc.clone().requires_grad_(True), w.clone().requires_grad_(True)
    out = DAF(f, ss.long(), st.long(), l2, w2)
    assert torch.equal(out, o32)
    out.backward(gout)
    assert f.grad.dtype == torch.bfloat16
    assert float((f.grad.float() - res["f32"][0]).abs().max() / res["f32"][0].abs().max()) < 5e-3      # one bf16 rounding
    assert torch.equal(l2.grad, res["f32"][1]) and torch.equal(w2.grad, res["f32"][2])
    # widths the 256-channel kernels do not cover are refused by the bf16 entries (the wrapper widens instead)
    from hipad_amd.lib import HipadError
    with pytest.raises(HipadError):
        lib.daf_forward(feat16[..., :128].contiguous(), ss, st, loc, w[..., :4].contiguous())


# ---- feature gradient of several call sites in one pass (hipad_daf_backward_feat_multi) -------------------------------
MULTI_SHAPES = [(16, 44), (8, 22), (4, 11), (2, 6)]


def _multi_calls(bs, specs, seed=70):
    """Calls (different anchors x points each) on ONE pyramid geometry: [(loc, w, gout)], plus the shared tables."""
    calls, ss, st, feat = [], None, None, None
    for k, (A, P) in enumerate(specs):
        f, ss, st, loc, w, gout = make_inputs(seed + k, bs, A, P, MULTI_SHAPES)
        feat = f if feat is None else feat
        calls.append((loc, w, gout))
    return feat, ss, st, calls


@pytest.mark.parametrize("bs", [1, 2])
def test_feat_multi_vs_cpu_oracle(lib, bs):
    """The frame's mix of call shapes (det / map / plan / ego point counts) through ONE sorted pass equals the sum of
    the oracle's per-call feature gradients; grad_feat is accumulated into (starts non-zero)."""
    feat, ss, st, calls = _multi_calls(bs, [(37, 13), (5, 300), (9, 90), (1, 13), (37, 13)])
    ref = np.zeros(feat.shape, np.float64)
    for loc, w, gout in calls:
        ref += O.daf_backward(feat.numpy(), ss, st, loc.numpy(), w.numpy(), gout.numpy(), acc64=True)[0]
    gf = torch.ones(feat.shape, device="cuda")
    lib.daf_backward_feat_multi([tuple(t.cuda() for t in c) for c in calls], gf, torch.from_numpy(ss).cuda(),
                                torch.from_numpy(st).cuda())
    assert rel_err(gf - 1, ref) < 1e-5


def test_feat_multi_equals_per_call_launches_and_chunks_long_tables(lib):
    """Same taps, same products as one hipad_daf_backward per call (only the order of the sums inside a row differs);
    more calls than HIPAD_DAF_MAX_CALLS are worked off in several passes."""
    feat, ss, st, calls = _multi_calls(1, [(7, 13), (3, 40)] * 35)     # 70 calls > 64
    assert len(calls) > lib.DAF_MAX_CALLS
    d_ss, d_st, d_feat = torch.from_numpy(ss).cuda(), torch.from_numpy(st).cuda(), feat.cuda()
    dc = [tuple(t.cuda() for t in c) for c in calls]
    one = torch.zeros_like(d_feat)
    for loc, w, gout in dc:
        lib.daf_backward(d_feat, d_ss, d_st, loc, w, gout, one, None, None)
    many = torch.zeros_like(d_feat)
    lib.daf_backward_feat_multi(dc, many, d_ss, d_st)
    assert rel_err(many, one.cpu().numpy()) < 1e-5


def test_feat_multi_through_the_autograd_sink(lib, monkeypatch):
    """shared_feature_grad + deferred feature gradient (the training step's path) == per-call pipelines."""
    from projects.mmdet3d_plugin.ops import deformable_aggregation as DA
    from projects.mmdet3d_plugin.ops import deformable_aggregation_function as DAF, shared_feature_grad
    feat, ss, st, calls = _multi_calls(2, [(37, 13), (5, 300), (9, 90), (1, 13)])
    d_ss, d_st = torch.from_numpy(ss).cuda().long(), torch.from_numpy(st).cuda().long()
    grads = {}
    for defer in (True, False):
        monkeypatch.setattr(DA, "DEFER_FEAT", defer)
        f = feat.cuda().to(torch.bfloat16).requires_grad_(True)      # the encoder's rows
        shared = shared_feature_grad(f)
        leaves, total = [], 0.0
        for loc, w, gout in calls:
            l, ww = loc.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
            leaves += [l, ww]
            total = total + (DAF(shared, d_ss, d_st, l, ww) * gout.cuda()).sum()
        total.backward()
        grads[defer] = [f.grad.float()] + [t.grad for t in leaves]
    assert rel_err(grads[True][0], grads[False][0].cpu().numpy()) < 1e-2      # bf16 leaf gradient: one rounding apart
    for a, b in zip(grads[True][1:], grads[False][1:]):
        assert torch.equal(a, b)                                             # grad_loc / grad_w: the same kernel


@pytest.mark.gpu
@pytest.mark.parametrize("run", [1, 2, 3, 16, 64])
def test_feat_multi_is_the_same_for_every_run_length(lib, run):
    """hipad_daf_set_feat_run only moves the boundaries between plain and atomic row updates: every run length gives
    the oracle's feature gradient (a run of 64 batches covers this whole case in a handful of waves)."""
    feat, ss, st, calls = _multi_calls(2, [(37, 13), (5, 300), (9, 90), (1, 13)])
    ref = np.zeros(feat.shape, np.float64)
    for loc, w, gout in calls:
        ref += O.daf_backward(feat.numpy(), ss, st, loc.numpy(), w.numpy(), gout.numpy(), acc64=True)[0]
    gf = torch.zeros(feat.shape, device="cuda")
    try:
        lib.load().hipad_daf_set_feat_run(run)
        lib.load().hipad_daf_set_feat_blocks(64 if run % 2 else 0)      # a small grid: every wave walks several runs
        lib.daf_backward_feat_multi([tuple(t.cuda() for t in c) for c in calls], gf, torch.from_numpy(ss).cuda(),
                                    torch.from_numpy(st).cuda())
    finally:
        lib.load().hipad_daf_set_feat_run(0)
        lib.load().hipad_daf_set_feat_blocks(0)
    assert rel_err(gf, ref) < 1e-5
