"""Image encoder (projects/mmdet3d_plugin/models/image_encoder.py: ResNet + FPN restated in plain torch with mmdet's
parameter names; the arithmetic of the reference lives in mmdet==2.28.2 / mmcv-full==1.7.1, which are not installed:
PARITY UNPINNED against mmdet itself).  What is pinned here:

  CPU  the checkpoint surface (mmdet's state_dict keys for ResNet50 / FPN), pretrained paths load or raise, the
       zero-initialised last BatchNorm of every bottleneck (mmdet's zero_init_residual default);
  GPU  the benchmarked configuration -- bf16 autocast, channels-last, MIOpen / CK kernels -- against the same modules run
       in fp32 on a seeded input: every pyramid level within 1e-2 of its largest magnitude (BASELINE.json bf16 class)."""
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
def test_bf16_channels_last_encoder_tracks_fp32():
    body, neck = build_encoder()
    fill_parameters_by_name(body, 11, scale=0.03)
    fill_parameters_by_name(neck, 12, scale=0.03)
    body, neck = body.cuda().train(), neck.cuda().train()      # training-mode BatchNorm (batch statistics), as in the step
    x = seeded((6, 3, 256, 704), 99).cuda()
    with torch.no_grad():
        ref = neck(body(x))
        xb = x.contiguous(memory_format=torch.channels_last)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            got = neck(body(xb))
    assert [tuple(t.shape) for t in got] == [(6, 256, 64, 176), (6, 256, 32, 88), (6, 256, 16, 44), (6, 256, 8, 22)]
    errs = [float((g.float() - r).abs().max() / r.abs().max()) for g, r in zip(got, ref)]
    rms = [float((g.float() - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt()) for g, r in zip(got, ref)]
    print("encoder bf16 vs fp32: max-rel per level", [round(e, 4) for e in errs], "rms-rel", [round(e, 4) for e in rms])
    assert all(e < 1e-2 for e in rms), rms
    assert all(e < 3e-2 for e in errs), errs     # 53 bf16 convolutions + batch statistics: worst element
