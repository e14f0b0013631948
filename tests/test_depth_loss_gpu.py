"""Dense-depth heads + loss on the flat pyramid (hip-ad_amd/csrc/depthloss.hip) against the module path
(projects/mmdet3d_plugin/models/blocks.py::DenseDepthNet: fp32 1x1 convolutions, exp, focal scaling, masked L1 in torch
operators -- itself pinned against the reference's DenseDepthNet in tests/test_losses.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-12))


def make_case(bs, cams, hws, seed, sparse=0.3):
    from projects.mmdet3d_plugin.models.blocks import DenseDepthNet
    g = torch.Generator().manual_seed(seed)
    net = DenseDepthNet(embed_dims=256, num_depth_layers=3, loss_weight=0.2).cuda()
    for m in net.depth_layers:
        torch.nn.init.normal_(m.weight, std=0.02, generator=None)
        torch.nn.init.constant_(m.bias, 1.5)
    blocks, off = [], 0
    for h, w in hws:
        blocks.append((off, cams * h * w))
        off += cams * h * w
    flat = (torch.randn(bs, off, 256, generator=g) * 2).to(torch.bfloat16).cuda()
    gts = []
    for h, w in hws[:3]:
        d = torch.rand(bs * cams, h, w, generator=g) * 70.0 + 0.5            # some beyond max_depth = 60
        gts.append(torch.where(torch.rand(bs * cams, h, w, generator=g) < sparse, d, torch.zeros_like(d)).cuda())
    focal = (torch.rand(bs, cams, generator=g) * 100 + 60).cuda()
    return net, flat, blocks, gts, focal


def levels_of(flat, blocks, cams, hws):
    bs = flat.shape[0]
    return [flat[:, o:o + n].view(bs, cams, h, w, 256).permute(0, 1, 4, 2, 3) for (o, n), (h, w) in zip(blocks, hws)]


@pytest.mark.parametrize("bs,cams,with_focal,shared", [(2, 3, True, True), (1, 6, False, True), (2, 2, True, False)])
def test_fused_depth_loss_matches_the_module_path(bs, cams, with_focal, shared):
    from projects.mmdet3d_plugin.ops import shared_feature_grad
    hws = [(8, 22), (4, 11), (2, 6), (1, 3)]
    net, flat, blocks, gts, focal = make_case(bs, cams, hws, seed=bs * 10 + cams)
    focal = focal if with_focal else None
    # module path: levels as views of a leaf pyramid, library convolutions, torch loss operators
    ref_flat = flat.clone().requires_grad_(True)
    ref = net.loss(net(levels_of(ref_flat, blocks, cams, hws), focal), gts)
    ref.backward()
    ref_grads = [(m.weight.grad.clone(), m.bias.grad.clone()) for m in net.depth_layers]
    for m in net.depth_layers:
        m.weight.grad = m.bias.grad = None
    # fused path on the same rows
    leaf = flat.clone().requires_grad_(True)
    feat = shared_feature_grad(leaf) if shared else leaf
    got = net.loss(net.on_pyramid(feat, blocks, cams, levels_of(feat, blocks, cams, hws), focal), gts)
    (got * 1.0).backward()
    assert abs(float(got) - float(ref)) <= 2e-5 * abs(float(ref)), (float(got), float(ref))
    # both feature gradients end as bf16 (the pyramid's dtype): equal to a rounding
    assert rel(leaf.grad.float(), ref_flat.grad.float()) < 6e-3
    touched = ref_flat.grad.float().abs().sum(-1) > 0
    assert torch.equal(leaf.grad.float().abs().sum(-1) > 0, touched) and 0 < int(touched.sum()) < touched.numel()
    for m, (gw, gb) in zip(net.depth_layers, ref_grads):
        assert rel(m.weight.grad, gw) < 1e-4 and rel(m.bias.grad, gb) < 1e-4


def test_fused_depth_loss_values_are_reproducible_and_scale_with_the_upstream_gradient():
    """The error sums are fixed-point integers: two evaluations give the same bits; d(3 * loss) = 3 * d(loss)."""
    hws = [(8, 22), (4, 11), (2, 6), (1, 3)]
    net, flat, blocks, gts, focal = make_case(2, 3, hws, seed=5)
    vals, grads = [], []
    for scale in (1.0, 1.0, 3.0):
        leaf = flat.clone().requires_grad_(True)
        loss = net.loss(net.on_pyramid(leaf, blocks, 3, None, focal), gts)
        for m in net.depth_layers:
            m.weight.grad = None
        (loss * scale).backward()
        vals.append(float(loss))
        grads.append((leaf.grad.float().clone(), net.depth_layers[0].weight.grad.clone()))
    assert vals[0] == vals[1] == vals[2]
    assert torch.equal(grads[0][0], grads[1][0])
    assert rel(grads[2][0], 3 * grads[0][0]) < 6e-3 and rel(grads[2][1], 3 * grads[0][1]) < 1e-5


def test_no_valid_target_gives_zero_loss_and_no_gradient():
    hws = [(8, 22), (4, 11), (2, 6), (1, 3)]
    net, flat, blocks, gts, focal = make_case(1, 2, hws, seed=7)
    gts = [torch.zeros_like(g) for g in gts]
    leaf = flat.clone().requires_grad_(True)
    loss = net.loss(net.on_pyramid(leaf, blocks, 2, None, focal), gts)
    loss.backward()
    assert float(loss) == 0.0 and float(leaf.grad.float().abs().max()) == 0.0


def test_bad_arguments_are_refused_before_any_launch():
    from hipad_amd import lib
    hws = [(8, 22), (4, 11), (2, 6), (1, 3)]
    net, flat, blocks, gts, focal = make_case(1, 2, hws, seed=8)
    w, b = net.depth_layers[0].weight.detach().reshape(-1), net.depth_layers[0].bias.detach()
    good = [(gts[0].reshape(-1), w, b, 8 * 22, 0)]
    lib.depth_loss_forward(flat, None, good, 2, 100.0, 60.0, 0.2)
    with pytest.raises(lib.HipadError):
        lib.depth_loss_forward(flat.float(), None, good, 2, 100.0, 60.0, 0.2)                      # not bf16 rows
    with pytest.raises(lib.HipadError):
        lib.depth_loss_forward(flat, None, [(gts[0].reshape(-1), w, b, 8 * 22, flat.shape[1])], 2, 100.0, 60.0, 0.2)   # rows outside
    with pytest.raises(lib.HipadError):
        lib.depth_loss_forward(flat, None, [(gts[1].reshape(-1), w, b, 8 * 22, 0)], 2, 100.0, 60.0, 0.2)   # gt of another level
    with pytest.raises(lib.HipadError):
        lib.depth_loss_forward(flat, focal.reshape(-1)[:1], good, 2, 100.0, 60.0, 0.2)             # focal too short
