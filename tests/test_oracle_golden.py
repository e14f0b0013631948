"""The CPU oracle (oracle/daf_oracle.c) against the golden vectors produced by the
reference's own PyTorch fallback (+ the CUDA kernel's border mask); see
tests/golden/make_golden.py.  This is what pins the oracle."""
import numpy as np
import pytest

from oracle import daf as O

CASES = ["daf_unit", "daf_multicam", "daf_ragged"]


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def off_kink(loc, spatial_shape, eps=1e-4):
    """Mask (bs,A,P,cams) of samples whose pixel coordinates are not within eps of an integer
    on any level.  Bilinear interpolation is continuous but not differentiable there: the
    kernel's floorf(loc*size-0.5) and grid_sample's floor(((2loc-1)+1)*size/2-0.5) may pick
    different sides, both valid one-sided derivatives -- grad_loc is compared off the kinks."""
    ok = np.ones(loc.shape[:-1], bool)
    cams = loc.shape[3]
    for c in range(cams):
        for h, w in spatial_shape[c]:
            for k, size in ((0, w), (1, h)):
                pix = loc[:, :, :, c, k].astype(np.float64) * size - 0.5
                ok[:, :, :, c] &= np.abs(pix - np.round(pix)) > eps
    return ok


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("acc64", [False, True])
def test_oracle_forward_matches_reference_fallback(golden, case, acc64):
    z = golden(case)
    out = O.daf_forward(z["feat"], z["spatial_shape"], z["scale_start_index"], z["loc"], z["weights"], acc64=acc64)
    # fp32 tolerance of the north star: 1e-3 relative; observed ~1e-6
    assert rel_err(out, z["out"]) < 2e-5


@pytest.mark.parametrize("case", CASES)
def test_oracle_backward_matches_reference_autograd(golden, case):
    z = golden(case)
    gf, gl, gw = O.daf_backward(z["feat"], z["spatial_shape"], z["scale_start_index"], z["loc"], z["weights"],
                                z["grad_out"], acc64=True)
    assert rel_err(gf, z["grad_feat"]) < 2e-5
    assert rel_err(gw, z["grad_weights"]) < 2e-5
    ok = off_kink(z["loc"], z["spatial_shape"])
    assert ok.mean() > 0.5  # the small ragged case is dominated by planted rim cases
    assert rel_err(gl[ok], z["grad_loc"][ok]) < 5e-5
    # masked samples: exactly zero gradient (deformable_aggregation_cuda.cu:232-235)
    loc = z["loc"]
    dropped = ~((loc[..., 0] > 0) & (loc[..., 0] < 1) & (loc[..., 1] > 0) & (loc[..., 1] < 1))
    assert dropped.any()
    assert np.all(gl[dropped] == 0) and np.all(gw[dropped] == 0)


def test_border_rule_differs_from_plain_grid_sample(golden):
    """The fixtures contain 'sliver' samples that grid_sample keeps and the kernel drops."""
    z = golden("daf_unit")
    loc = z["loc"]
    w = z["spatial_shape"][0, 0, 1]
    sliver = ((loc[..., 0] <= 0) & (loc[..., 0] > -0.5 / w)) | ((loc[..., 0] >= 1) & (loc[..., 0] < 1 + 0.5 / w))
    assert sliver.any()


def test_taps_index_work(golden):
    z = golden("daf_multicam")
    valid, taps = O.daf_taps(z["spatial_shape"], z["scale_start_index"], z["loc"], z["feat"].shape[1])
    loc = z["loc"]
    expect = (loc[..., 0] > 0) & (loc[..., 0] < 1) & (loc[..., 1] > 0) & (loc[..., 1] < 1)
    assert np.array_equal(valid.astype(bool), expect)
    # integer corner of a valid sample = floor(loc*size-0.5) in fp32
    ss = z["spatial_shape"].astype(np.int32)
    b, a, p, c = np.argwhere(expect)[0]
    for s in range(ss.shape[1]):
        h, w_ = ss[c, s]
        hl = int(np.floor(np.float32(np.float32(loc[b, a, p, c, 1]) * np.float32(h)) - np.float32(0.5)))
        wl = int(np.floor(np.float32(np.float32(loc[b, a, p, c, 0]) * np.float32(w_)) - np.float32(0.5)))
        assert taps[b, a, p, c, s, 0] == hl and taps[b, a, p, c, s, 1] == wl


def test_threaded_oracle_is_bitwise_the_sequential_one():
    """bench.py's cpu_baseline deals anchors to several host threads (oracle/daf.py set_threads); the results must be
    bit for bit the one-thread checker's: an anchor's output row is summed by one thread in the sequential order, and the
    feature-gradient scatter is partitioned by pyramid row, each row receiving its addends in the sequential order."""
    import numpy as np
    from oracle import daf as O
    rng = np.random.default_rng(3)
    cams, A, P = 3, 41, 7
    shapes = [(8, 22), (4, 11)]
    ss = np.array([shapes] * cams, np.int32)
    sizes = (ss[..., 0] * ss[..., 1]).reshape(-1)
    st = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32).reshape(cams, -1)
    feat = rng.standard_normal((2, int(sizes.sum()), 64), dtype=np.float32)
    loc = rng.random((2, A, P, cams, 2), dtype=np.float32) * 1.4 - 0.2
    w = rng.random((2, A, P, cams, 2, 8), dtype=np.float32)
    go = rng.standard_normal((2, A, 64), dtype=np.float32)
    try:
        O.set_threads(1)
        ref = (O.daf_forward(feat, ss, st, loc, w),) + tuple(O.daf_backward(feat, ss, st, loc, w, go))
        ref64 = O.daf_backward(feat, ss, st, loc, w, go, acc64=True)
        O.set_threads(4)
        got = (O.daf_forward(feat, ss, st, loc, w),) + tuple(O.daf_backward(feat, ss, st, loc, w, go))
        got64 = O.daf_backward(feat, ss, st, loc, w, go, acc64=True)
    finally:
        O.set_threads(1)
    assert all(np.array_equal(a, b) for a, b in zip(ref, got))
    assert all(np.array_equal(a, b) for a, b in zip(ref64, got64))
