"""LayerNorm kernel (hip-ad_amd/csrc/layernorm.hip) against torch.nn.functional.layer_norm in fp32:
forward, input gradient, gamma / beta gradients accumulated on top of existing .grad buffers."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a.detach() - b.detach()).abs().max() / b.detach().abs().max().clamp_min(1e-12))


@pytest.mark.parametrize("M,N", [(1481, 256), (1481, 512), (900, 32), (5400, 256), (1, 256), (7, 64), (333, 128),
                                 (100, 1024), (48, 768), (3, 4)])
def test_layer_norm_forward_backward(M, N):
    from hipad_amd import functional as HF
    g = torch.Generator().manual_seed(M * 31 + N)
    x = (torch.randn(M, N, generator=g) * 3 + 0.5).cuda()
    w = (torch.rand(N, generator=g) + 0.5).cuda()
    b = torch.randn(N, generator=g).cuda()
    go = torch.randn(M, N, generator=g).cuda()
    x1 = x.clone().requires_grad_(True)
    w1, b1 = torch.nn.Parameter(w.clone()), torch.nn.Parameter(b.clone())
    w1.grad, b1.grad = torch.full_like(w1, 0.5), torch.full_like(b1, -0.25)  # accumulate on top of these
    y = HF.layer_norm(x1, w1, b1, 1e-5)
    y.backward(go)
    x2 = x.clone().requires_grad_(True)
    w2, b2 = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(x2, (N,), w2, b2, 1e-5)
    yr.backward(go)
    assert rel(y, yr) < 1e-5
    assert rel(x1.grad, x2.grad) < 2e-5
    assert rel(w1.grad - 0.5, w2.grad) < 2e-5
    assert rel(b1.grad + 0.25, b2.grad) < 2e-5


def test_layer_norm_module_3d_and_fresh_grads():
    from hipad_amd.compat import LayerNorm
    ln = LayerNorm(256).cuda()
    ref = torch.nn.LayerNorm(256).cuda()
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5); ln.bias.normal_()
    ref.load_state_dict(ln.state_dict())
    x = torch.randn(2, 37, 256).cuda()
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ln(xa).square().sum().backward()
    ref(xb).square().sum().backward()
    assert rel(xa.grad, xb.grad) < 2e-5
    assert rel(ln.weight.grad, ref.weight.grad) < 2e-5 and rel(ln.bias.grad, ref.bias.grad) < 2e-5
