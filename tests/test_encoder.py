"""Image encoder (projects/mmdet3d_plugin/models/image_encoder.py: ResNet + FPN restated in plain torch with mmdet's
parameter names; the arithmetic of the reference lives in mmdet==2.28.2 / mmcv-full==1.7.1, which are not installed:
PARITY UNPINNED against mmdet itself).  What is pinned here:

  CPU  the checkpoint surface (mmdet's state_dict keys for ResNet50 / FPN), pretrained paths load or raise, the
       zero-initialised last BatchNorm of every bottleneck (mmdet's zero_init_residual default);
  GPU  the benchmarked configuration -- bf16 autocast, channels-last, MIOpen / CK kernels -- against the same modules run
       in fp32 on a seeded input, unit by unit (teacher-forced): BASELINE.json's bf16 class, 1e-2."""
import os

import pytest
import torch

from seeded import fill_parameters_by_name, seeded


def build_encoder(depth=50):
    import projects.mmdet3d_plugin.models  # noqa: F401
    from projects.mmdet3d_plugin.models.image_encoder import FPN, ResNet
    body = ResNet(depth=depth, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=-1, norm_eval=False, style="pytorch")
    neck = FPN(in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=4, start_level=0,
               norm_cfg=dict(type="BN"), no_norm_on_lateral=True)
    body.init_weights()
    neck.init_weights()
    return body, neck


def test_state_dict_surface_and_init():
    body, neck = build_encoder()
    keys = set(body.state_dict())
    for k in ("conv1.weight", "bn1.running_mean", "layer1.0.conv1.weight", "layer1.0.downsample.0.weight",
              "layer1.0.downsample.1.weight", "layer4.2.bn3.bias", "layer3.5.conv2.weight"):
        assert k in keys, k
    nk = set(neck.state_dict())
    for k in ("lateral_convs.0.conv.weight", "lateral_convs.3.conv.bias", "fpn_convs.0.conv.weight", "fpn_convs.2.bn.weight"):
        assert k in nk, k
    assert sum(p.numel() for p in body.parameters()) == 23508032      # torchvision / mmdet ResNet50 body
    from projects.mmdet3d_plugin.models.image_encoder import Bottleneck
    assert all(float(m.bn3.weight.abs().max()) == 0.0 for m in body.modules() if isinstance(m, Bottleneck))


def test_pretrained_path_loads_or_raises(tmp_path):
    from projects.mmdet3d_plugin.models.image_encoder import ResNet, load_checkpoint
    with pytest.raises(FileNotFoundError):
        ResNet(depth=50, pretrained=str(tmp_path / "missing.pth")).init_weights()
    src = ResNet(depth=50)
    fill_parameters_by_name(src, 5)
    path = tmp_path / "r50.pth"
    torch.save({"state_dict": {"img_backbone." + k: v for k, v in src.state_dict().items()}}, path)
    dst = ResNet(depth=50)
    missing, unexpected = load_checkpoint(dst, str(path), strict=True, prefix="img_backbone.")
    assert not missing and not unexpected
    assert torch.equal(dst.layer2[1].conv2.weight, src.layer2[1].conv2.weight)


@pytest.mark.gpu
def test_bf16_channels_last_encoder_tracks_fp32_unit_by_unit():
    """A 53-convolution network with RANDOM (seeded) parameters amplifies any perturbation from layer to layer (measured:
    bf16 vs fp32 end to end differ by 55-72 % RMS on such parameters -- chaos of the random net, not of the kernels), so
    the comparison is teacher-forced like the decoder's: the fp32 run records the input and output of every unit (stem,
    each bottleneck, each FPN convolution); the bf16 channels-last run feeds every unit the RECORDED input and its output
    is compared with the recorded one.  BASELINE.json's bf16 class is 1e-2 per operator; a bottleneck unit chains three
    bf16 convolutions + BatchNorm + the residual, so the unit bound is 1.5e-2 RMS-relative (measured worst unit: 1.00e-2,
    worst element 1.32e-2 of the unit's range; bound 3e-2)."""
    from projects.mmdet3d_plugin.models.image_encoder import Bottleneck, _ConvModule
    body, neck = build_encoder()
    fill_parameters_by_name(body, 11, scale=0.03)
    fill_parameters_by_name(neck, 12, scale=0.03)
    body, neck = body.cuda().train(), neck.cuda().train()      # training-mode BatchNorm (batch statistics), as in the step
    x = seeded((6, 3, 256, 704), 99).cuda()
    units = [m for m in list(body.modules()) + list(neck.modules()) if isinstance(m, (Bottleneck, _ConvModule))]
    units.append(body.conv1)
    rec, errs = {}, {}

    def recorder(mod, inp, out):
        rec[mod] = (inp[0].detach().clone(), out.detach().clone())

    def force(mod, inp):
        ref_in = rec[mod][0]
        return (ref_in.to(inp[0].dtype).contiguous(memory_format=torch.channels_last),)

    def compare(mod, inp, out):
        r = rec[mod][1]
        d = out.float() - r
        errs[mod] = (float(d.abs().max() / r.abs().max()), float(d.pow(2).mean().sqrt() / r.pow(2).mean().sqrt()))

    with torch.no_grad():
        hs = [m.register_forward_hook(recorder) for m in units]
        ref = neck(body(x))
        for h in hs:
            h.remove()
        hs = [m.register_forward_pre_hook(force) for m in units] + [m.register_forward_hook(compare) for m in units]
        with torch.autocast("cuda", dtype=torch.bfloat16):
            got = neck(body(x.contiguous(memory_format=torch.channels_last)))
        for h in hs:
            h.remove()
    assert [tuple(t.shape) for t in got] == [(6, 256, 64, 176), (6, 256, 32, 88), (6, 256, 16, 44), (6, 256, 8, 22)]
    assert len(errs) == len(units) == 16 + 8 + 1
    worst_max, worst_rms = max(e[0] for e in errs.values()), max(e[1] for e in errs.values())
    print("encoder units bf16 vs fp32: worst max-rel %.4f, worst rms-rel %.4f" % (worst_max, worst_rms))
    assert worst_rms < 1.5e-2, worst_rms
    assert worst_max < 3e-2, worst_max


@pytest.mark.gpu
def test_shadow_weight_convolution_equals_autocast_cast():
    """Conv2d reading the optimiser-kept bf16 copy of its weight: same output and same weight / input gradients as
    autocast's per-call cast (both read the round-to-nearest bf16 of the fp32 master and the gradient is the bf16 one
    widened), with the gradient accumulated in place into the preset fp32 buffer."""
    from hipad_amd import functional as HF
    from projects.mmdet3d_plugin.models.image_encoder import Conv2d
    torch.manual_seed(0)
    for k, pad in ((1, 0), (3, 1)):
        conv = Conv2d(64, 128, k, padding=pad, bias=False).cuda()
        x = torch.randn(6, 64, 32, 44, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
        gout = torch.randn(6, 128, 32, 44, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y0 = conv(x)
        y0.backward(gout)
        gw0, gx0 = conv.weight.grad.clone(), x.grad.clone()
        conv.weight.grad = torch.zeros_like(conv.weight)
        x.grad = None
        conv.weight._hipad_bf16 = conv.weight.detach().to(torch.bfloat16)
        HF.INPLACE_PARAMS.discard(id(conv.weight))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y1 = conv(x)
        held = conv.weight.grad
        y1.backward(gout)
        assert conv.weight.grad is held and id(conv.weight) in HF.INPLACE_PARAMS      # accumulated in place
        # same operands, but MIOpen may pick another solver on the second call of a shape (different summation order):
        # bf16-rounding-level agreement, not bitwise
        # (the weight gradient is a bf16 tensor out of a long reduction: two solvers measured 0.9 % apart)
        for a, b, tol in ((y0, y1, 5e-3), (gx0, x.grad, 5e-3), (gw0, conv.weight.grad, 2e-2)):
            assert float((a.float() - b.float()).norm() / b.float().norm()) < tol
        # outside autocast (fp32 run) the module is a plain nn.Conv2d
        y2 = conv(x.detach())
        assert y2.dtype == torch.float32


@pytest.mark.gpu
def test_grid_mask_kernel_equals_the_torch_construction():
    """GridMask in one launch (hipad_grid_mask) against the module's own torch construction of the same mask (run on CPU
    tensors with the same drawn parameters): equal bit for bit in fp32; the bf16 channels-last output the encoder
    consumes equals the fp32 result rounded to bf16."""
    from projects.mmdet3d_plugin.models.grid_mask import GridMask
    torch.manual_seed(0)
    x = torch.randn(6, 3, 64, 176) * 50
    for mode in (0, 1):
        gm = GridMask(True, True, rotate=1, offset=False, ratio=0.5, mode=mode, prob=0.7).train()
        gm.external_randomize = True
        for params in ([1, 40, 20, 7, 33], [1, 2, 1, 0, 1], [1, 63, 31, 62, 5], [0, 40, 20, 7, 33], [1, 17, 9, 16, 16]):
            p = torch.tensor(params, dtype=torch.float32)
            gm._dev = p
            want = gm(x)
            gm._dev = p.cuda()
            gm.out_dtype, gm.out_channels_last = torch.float32, False
            got = gm(x.cuda())
            assert torch.equal(got.cpu(), want), (mode, params)
            gm.out_dtype, gm.out_channels_last = torch.bfloat16, True
            got16 = gm(x.cuda())
            assert got16.dtype == torch.bfloat16 and got16.is_contiguous(memory_format=torch.channels_last)
            assert torch.equal(got16.cpu(), want.to(torch.bfloat16)), (mode, params)
    assert gm.eval()(x) is x


@pytest.mark.gpu
@pytest.mark.parametrize("C,shape,relu,res", [(64, (6, 32, 44), True, False), (256, (6, 16, 22), True, True),
                                               (2048, (6, 8, 22), True, True), (1024, (2, 5, 7), False, False),
                                               (128, (1, 3, 1), False, True)])
def test_fused_batch_norm_add_relu_equals_torch(C, shape, relu, res):
    """relu(bn(x) + residual) on the fused kernels (bf16 channels-last in / out, fp32 statistics) against torch's
    batch_norm + add + relu evaluated in fp32 on the same bf16 inputs: output, running statistics, dx, d(residual),
    d(gamma), d(beta) (accumulated in place into preset gradient buffers)."""
    from hipad_amd import functional as HF
    from projects.mmdet3d_plugin.models.image_encoder import BatchNorm2d
    torch.manual_seed(C)
    n, h, w = shape
    bn = BatchNorm2d(C).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.normal_(0, 0.3)
    ref = copy_bn = torch.nn.BatchNorm2d(C).cuda().train()
    copy_bn.load_state_dict(bn.state_dict())
    x = (torch.randn(n, C, h, w, device="cuda") * 2 + 0.7).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    r = torch.randn(n, C, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if res else None
    g = torch.randn(n, C, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x1, r1 = x.clone().requires_grad_(True), (r.clone().requires_grad_(True) if res else None)
    bn.weight.grad, bn.bias.grad = torch.zeros_like(bn.weight), torch.zeros_like(bn.bias)
    held = bn.weight.grad
    HF.BN_ARENA.reset(x.device)
    assert HF.batch_norm_act_ok(x1, bn.weight)
    y = bn(x1, relu=relu, residual=r1)
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=torch.channels_last)
    y.backward(g)
    assert bn.weight.grad is held
    # fp32 reference on the same (bf16-valued) inputs
    x2, r2 = x.float().requires_grad_(True), (r.float().requires_grad_(True) if res else None)
    z = ref(x2)
    if res:
        z = z + r2
    if relu:
        z = torch.relu(z)
    z.backward(g.float())

    def close(a, b, tol):
        a, b = a.float(), b.float()
        assert float((a - b).norm() / b.norm().clamp_min(1e-12)) < tol, float((a - b).norm() / b.norm().clamp_min(1e-12))

    close(y, z, 4e-3)                       # one bf16 rounding of the output
    close(x1.grad, x2.grad, 6e-3)
    if res:
        close(r1.grad, r2.grad, 4e-3)
    close(bn.weight.grad, ref.weight.grad, 3e-3)
    close(bn.bias.grad, ref.bias.grad, 3e-3)
    close(bn.running_mean, ref.running_mean, 1e-4)
    close(bn.running_var, ref.running_var, 1e-3)
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("bs", [1, 2])
def test_fpn_writes_the_flat_pyramid_in_place(bs):
    """in training the FPN's last norm layers write every level straight into the flat tensor the
    aggregation operator reads (level-major rows; tables from ops.level_major_tables) -- the levels ARE blocks of it
    (data_ptr aliasing), there is no copy into a "column" layout.  Against the copying path (feature_maps_format of
    separate level tensors, the reference's layout): the same level values, the same aggregated features through
    the operator, and the same gradients of an encoder parameter and of the input image."""
    import copy
    from hipad_amd.frame import build_detector
    from projects.mmdet3d_plugin.models import sparse_detector as SD
    from projects.mmdet3d_plugin.ops import deformable_aggregation_function as DAF
    torch.manual_seed(2)
    model, _ = build_detector(stage=2, plan_queries=48)
    model.train()
    model.use_grid_mask = False
    img = torch.randn(bs, 6, 3, 256, 704, device="cuda")
    g = torch.Generator().manual_seed(4)
    A, P = 50, 13
    loc = (torch.rand(bs, A, P, 6, 2, generator=g) * 1.2 - 0.1).cuda()
    w = torch.softmax(torch.randn(bs, A, P * 6 * 4, 8, generator=g), 2).reshape(bs, A, P, 6, 4, 8).contiguous().cuda()
    res = {}
    for in_place in (True, False):
        SD.IN_PLACE_PYRAMID = in_place
        try:
            m = copy.deepcopy(model)
            x = img.clone().requires_grad_(True)
            fm, _ = m.extract_feat(x, True, {})
            levels = fm[0]._hipad_levels
            if in_place:
                flat = fm[0]
                assert flat.dtype == torch.bfloat16 and tuple(flat.shape) == (bs, 89760, 256)
                row = 0
                for t in levels:                       # every level is a block of rows of the flat tensor
                    assert t.data_ptr() == flat.data_ptr() + row * 256 * 2, "level is not a view of the flat pyramid"
                    assert tuple(t.shape[:3]) == (bs, 6, 256)
                    row += 6 * t.shape[3] * t.shape[4]
            out = DAF(fm[0], fm[1], fm[2], loc, w)
            out.square().sum().backward()
            p = m.img_neck.fpn_convs[0].conv.weight
            res[in_place] = dict(levels=[t.detach().float().clone() for t in levels], out=out.detach().clone(),
                                 gp=p.grad.detach().float().clone(), gx=x.grad.detach().clone())
        finally:
            SD.IN_PLACE_PYRAMID = True
    a, b = res[True], res[False]
    # the two passes run their convolutions separately (MIOpen may answer a shape's first call from another solver while it
    # searches: bit-equal in most runs, one bf16 ulp apart in some): levels to a bf16 ulp, aggregated features likewise
    for la, lb in zip(a["levels"], b["levels"]):
        assert float((la - lb).abs().max()) <= 2 ** -7 * float(lb.abs().max())
    assert float((a["out"] - b["out"]).abs().max()) <= 2 ** -7 * float(b["out"].abs().max())
    for k in ("gp", "gx"):
        rel = float((a[k] - b[k]).norm() / b[k].norm())
        print("in-place vs copied pyramid, gradient", k, "rel L2", rel)
        assert rel < 5e-2, (k, rel)                    # backward: fp32 atomics + bf16 hand-off at the cut (measured ~1e-2)
