"""Attention kernel (hip-ad_amd/csrc/attn.hip) against a plain PyTorch fp32 reference of the same op,
softmax(q k^T / sqrt(D)) v per head.  The reference's arithmetic for this block lives in
flash-attn==2.7.0.post2 (not vendored, CUDA-only): parity for it is unpinned by the reference's own
files; the published algorithm is restated here in torch.  Tolerance: 1e-2 relative (bf16 operands),
BASELINE.json north_star."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-2


def ref_attention(q, k, v, heads, mask_keep=None, inv_keep=1.0):
    B, Nq, E = q.shape
    D = E // heads
    qh = q.view(B, Nq, heads, D).transpose(1, 2)
    kh = k.view(B, -1, heads, D).transpose(1, 2)
    vh = v.view(B, -1, heads, D).transpose(1, 2)
    p = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(D), dim=-1)
    if mask_keep is not None:
        p = p * mask_keep * inv_keep
    return (p @ vh).transpose(1, 2).reshape(B, Nq, E)


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


# decoder shapes: det self 900x900 D64, temporal det 900x600 D64, map 100x100 D32,
# plan+ego -> det+map 481x1000 D32, ragged tails
SHAPES = [(1, 900, 900, 64), (2, 900, 600, 64), (1, 100, 100, 32), (1, 481, 1000, 32), (2, 481, 600, 32),
          (1, 1, 7, 32), (1, 17, 33, 64), (1, 50, 31, 128)]


@pytest.mark.parametrize("B,Nq,Nk,D", SHAPES)
def test_forward_backward_vs_torch_fp32(B, Nq, Nk, D):
    from hipad_amd import functional as HF
    H = 8
    g = torch.Generator().manual_seed(Nq * 7 + Nk)
    q, k, v = (torch.randn(B, n, H * D, generator=g).cuda() for n in (Nq, Nk, Nk))
    go = torch.randn(B, Nq, H * D, generator=g).cuda()
    q1, k1, v1 = (t.clone().requires_grad_(True) for t in (q, k, v))
    out = HF.attention(q1, k1, v1, H)
    out.backward(go)
    q2, k2, v2 = (t.clone().requires_grad_(True) for t in (q, k, v))
    ref = ref_attention(q2, k2, v2, H)
    ref.backward(go)
    assert rel(out, ref) < TOL
    assert rel(q1.grad, q2.grad) < 2 * TOL
    assert rel(k1.grad, k2.grad) < 2 * TOL
    assert rel(v1.grad, v2.grad) < 2 * TOL


def test_matches_bf16_rounded_reference_tightly():
    """Against a reference that rounds q, k, v to bf16 first, the kernel agrees to ~1e-3: what is left
    is P rounded to bf16 before the second product."""
    from hipad_amd import functional as HF
    H, D = 8, 64
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(1, n, H * D, generator=g).cuda() for n in (300, 260, 260))
    out = HF.attention(q, k, v, H)
    ref = ref_attention(*(t.bfloat16().float() for t in (q, k, v)), H)
    assert rel(out, ref) < 4e-3


def test_softmax_spike_and_large_scores():
    """One key dominates one query by a wide margin (running-max update path) and scores are large."""
    from hipad_amd import functional as HF
    H, D = 8, 32
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(1, n, H * D, generator=g).cuda() * 3 for n in (64, 200, 200))
    k[0, 150] = q[0, 10] * 4  # spike late in the key sequence
    out = HF.attention(q, k, v, H)
    ref = ref_attention(q.bfloat16().float(), k.bfloat16().float(), v.bfloat16().float(), H)
    assert torch.isfinite(out).all()
    assert rel(out, ref) < TOL


def test_dropout_is_consistent_between_forward_and_backward():
    """With dropout the op is out = (P o M / (1-p)) V for a fixed mask M(seed).  Recover M from a
    forward with v = identity-like probes, then check forward and all three gradients against torch
    using that very mask."""
    from hipad_amd import functional as HF
    B, H, D, Nq, Nk, pd, seed = 1, 8, 32, 48, 32, 0.25, 1234
    g = torch.Generator().manual_seed(9)
    q, k = (torch.randn(B, n, H * D, generator=g).cuda() for n in (Nq, Nk))
    # probe: v[key, h*D + d] = 1 if d == key (Nk == D) -> out[q, h, d] = dropped P[q, key=d]
    v_probe = torch.eye(Nk, D).repeat(1, H).reshape(1, Nk, H * D).cuda()
    pd_out = HF.attention(q, k, v_probe, H, p_drop=pd, seed=seed).view(B, Nq, H, D).transpose(1, 2)  # (B,H,Nq,Nk)
    keep = (pd_out > 0).float()
    frac = float(keep.mean())
    assert abs(frac - (1 - pd)) < 0.03, frac
    # same seed again -> same mask
    again = HF.attention(q, k, v_probe, H, p_drop=pd, seed=seed).view(B, Nq, H, D).transpose(1, 2)
    assert torch.equal(again > 0, pd_out > 0)
    other = HF.attention(q, k, v_probe, H, p_drop=pd, seed=seed + 1).view(B, Nq, H, D).transpose(1, 2)
    assert not torch.equal(other > 0, pd_out > 0)
    v = torch.randn(B, Nk, H * D, generator=g).cuda()
    go = torch.randn(B, Nq, H * D, generator=g).cuda()
    q1, k1, v1 = (t.clone().requires_grad_(True) for t in (q, k, v))
    out = HF.attention(q1, k1, v1, H, p_drop=pd, seed=seed)
    out.backward(go)
    q2, k2, v2 = (t.clone().requires_grad_(True) for t in (q, k, v))
    ref = ref_attention(q2, k2, v2, H, mask_keep=keep, inv_keep=1 / (1 - pd))
    ref.backward(go)
    assert rel(out, ref) < TOL
    assert rel(q1.grad, q2.grad) < 2 * TOL
    assert rel(k1.grad, k2.grad) < 2 * TOL
    assert rel(v1.grad, v2.grad) < 2 * TOL


def test_bad_head_dim_is_rejected():
    from hipad_amd import functional as HF, lib
    q = torch.randn(1, 4, 8 * 40).cuda()
    with pytest.raises(lib.HipadError):
        HF.attention(q, q, q, 8)
