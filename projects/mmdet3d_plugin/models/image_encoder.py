"""ResNet + FPN image encoder in plain torch (MIOpen convolutions), with mmdet's parameter names.

The reference takes these two modules from mmdet==2.28.2 (``type="ResNet"`` / ``type="FPN"`` in
projects/configs/hipad_b2d_stage2.py:112-134; built at models/sparse_detector.py:45-47), a
third-party package that is not installed here.  This file restates the published architecture for
the options those configs use -- bottleneck ResNet (depth 50/101, ``style="pytorch"``: stride on the
3x3 conv), BatchNorm, optional activation checkpointing per stage block (``with_cp``); FPN with 1x1
laterals without norm, nearest-neighbour top-down path, 3x3 output convs with BatchNorm -- and keeps
mmdet's module/parameter names (``layer1.0.conv1.weight``, ``lateral_convs.0.conv.weight``,
``fpn_convs.0.bn.weight`` ...) so ImageNet / HiP-AD checkpoints load.  Parity for it is unpinned by the
reference's own files; tests check shapes, names and gradient flow.
When mmdet is importable its registries already hold the real classes and these are not registered.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint as cp

from hipad_amd.compat import BACKBONES, HAVE_MMCV, NECKS, BaseModule

__all__ = ["ResNet", "FPN"]

_DEPTH = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


def load_checkpoint(module, path, strict=False, prefix=None):
    """state_dict loader restricted to ``torch.load(..., weights_only=True)`` (nothing from the file is executed).
    Accepts a bare state_dict or mmcv's ``{"state_dict": ...}`` wrapper; ``prefix`` strips e.g. "img_backbone.".
    Raises FileNotFoundError when the file is absent -- there is no silent fall back to random weights."""
    import os
    if not os.path.exists(path):
        raise FileNotFoundError(f"checkpoint {path!r} requested by the config does not exist (set it to None for a "
                                "random-init run)")
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    if prefix:
        sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    missing, unexpected = module.load_state_dict(sd, strict=strict)
    return missing, unexpected


import os as _os

from hipad_amd import functional as HF

# BatchNorm + identity add + ReLU on the fused kernels (csrc/batchnorm.hip) for bf16 channels-last training inputs
USE_FUSED_BN = _os.environ.get("HIPAD_FUSED_BN", "1") == "1"


class _ShadowWeight(torch.autograd.Function):
    """weight (fp32 master) -> its bf16 copy kept current by the optimiser kernel: no cast launch forward; backward adds
    the bf16 weight gradient straight into the fp32 ``weight.grad`` buffer (one launch, where autocast's cast needs one
    to widen the gradient and AccumulateGrad another to re-layout a channels-last one)."""

    @staticmethod
    def forward(ctx, weight, shadow):
        ctx.weight = weight
        return shadow.view(weight.shape)

    @staticmethod
    def backward(ctx, g):
        from hipad_amd import functional as HF
        w = ctx.weight
        tgt = w.grad if (HF.LINEAR_INPLACE_GRAD and w.is_leaf) else None
        if tgt is not None and tgt.dtype == torch.float32 and tgt.is_contiguous():
            HF.INPLACE_PARAMS.add(id(w))
            if BATCH_WEIGHT_GRADS and g.is_cuda and g.dtype == torch.bfloat16 and g.dim() <= 4:
                # all convolutions' gradients of this backward pass go into their fp32 buffers in ONE launch when the
                # pass ends (engine callback), instead of one add per layer
                task = torch._C._current_graph_task_id()
                if _PASS[0] != task:            # first convolution of this backward pass
                    _PASS[0] = task
                    _PENDING.clear()            # (leftovers of a pass that ended in an exception)
                    torch.autograd.Variable._execution_engine.queue_callback(_flush_weight_grads)
                _PENDING.append((tgt, g))
            else:
                tgt.add_(g)
            return None, None
        return g.to(w.dtype), None


BATCH_WEIGHT_GRADS = _os.environ.get("HIPAD_BATCH_WEIGHT_GRADS", "1") == "1"
_PENDING = []
_PASS = [None]


def _flush_weight_grads():
    from hipad_amd import lib as _lib
    pairs = list(_PENDING)
    _PENDING.clear()
    _PASS[0] = None
    by_dev = {}
    for tgt, g in pairs:
        by_dev.setdefault(tgt.device, []).append((tgt, g))
    for dev_pairs in by_dev.values():
        try:
            _lib.accumulate_bf16(dev_pairs)
        except _lib.HipadLayoutError:
            # a gradient layout the table kernel does not take (found on the host before any launch): one add per
            # tensor, as without batching
            for tgt, g in dev_pairs:
                tgt.add_(g)


class Conv2d(nn.Conv2d):
    """nn.Conv2d (same parameters / state_dict).  When its weight carries ``_hipad_bf16`` -- the bf16 copy the flat AdamW
    kernel rewrites with every update (hipad_amd.frame.TrainStep attaches it) -- and the call runs under bf16 autocast,
    the convolution reads that copy instead of casting the fp32 master again each frame (80 cast launches per frame)."""

    def forward(self, x):
        shadow = getattr(self.weight, "_hipad_bf16", None)
        if shadow is not None:
            from hipad_amd.optim import shadow_is_current
            if not shadow_is_current(self.weight):      # written by torch since the copy was made and nobody to refresh it
                shadow = None
        if (shadow is not None and x.is_cuda and torch.is_autocast_enabled() and self.padding_mode == "zeros"
                and torch.get_autocast_dtype("cuda") == torch.bfloat16):
            if torch.is_grad_enabled() and self.weight.requires_grad:
                w = _ShadowWeight.apply(self.weight, shadow)
            else:
                w = shadow.view(self.weight.shape)
            return nn.functional.conv2d(x, w, self.bias, self.stride, self.padding, self.dilation, self.groups)
        return super().forward(x)


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d (same parameters, buffers and state_dict) whose ``num_batches_tracked += 1`` can be deferred: 61
    norm layers bump a one-element int64 counter each, 61 single-block launches per frame.  With ``defer_counter`` set
    (SparseDetector does, for the modules it owns) the layer only queues its counter and ``flush_counters()`` -- called
    once after the neck -- adds one to all of them in one multi-tensor launch.  The counter does not enter the
    arithmetic (momentum is a number, not None)."""
    defer_counter = False
    _pending = []

    def fused_path(self, x):
        """True when ``forward(x)`` runs on the two fused kernels (training statistics, bf16 channels-last, supported width)."""
        return bool(self.training and self.track_running_stats and self.momentum is not None and self.affine and USE_FUSED_BN
                    and HF.batch_norm_act_ok(x, self.weight))

    def forward(self, x, relu=False, residual=None, out=None):
        """``relu`` / ``residual`` (ours): relu?(bn(x) (+ residual)) -- on bf16 channels-last training inputs the whole
        expression is two launches (hipad_bn_forward), two more in the backward; otherwise the torch ops.
        ``out`` (fused path only): (samples, rows, C) block of the flat pyramid the result is written into; the result
        then comes back as the (samples, cameras, C, h, w) level."""
        deferred = self.defer_counter and self.training and self.track_running_stats and self.momentum is not None
        if (self.training and self.track_running_stats and self.momentum is not None and self.affine and USE_FUSED_BN
                and HF.batch_norm_act_ok(x, self.weight)):
            if deferred:
                BatchNorm2d._pending.append(self.num_batches_tracked)
            else:
                self.num_batches_tracked.add_(1)
            return HF.batch_norm_act(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps, self.momentum,
                                     relu, residual, out)
        if out is not None:
            raise ValueError("BatchNorm2d: `out` needs the fused path (check fused_path(x) first)")
        if deferred:
            BatchNorm2d._pending.append(self.num_batches_tracked)
            y = nn.functional.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, True,
                                         self.momentum, self.eps)
        else:
            y = super().forward(x)
        if residual is not None:
            y = y + residual
        return nn.functional.relu(y) if relu else y

    @staticmethod
    def flush_counters():
        pend, BatchNorm2d._pending = BatchNorm2d._pending, []
        if pend:
            torch._foreach_add_(pend, 1)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, with_cp=False):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.with_cp = with_cp

    def _body(self, x):
        out = self.bn1(self.conv1(x), relu=True)
        out = self.bn2(self.conv2(out), relu=True)
        identity = x if self.downsample is None else self.downsample(x)
        return self.bn3(self.conv3(out), relu=True, residual=identity)      # relu(bn3(.) + identity)

    def forward(self, x):
        if self.with_cp and x.requires_grad:
            return cp.checkpoint(self._body, x, use_reentrant=False)
        return self._body(x)


class ResNet(BaseModule):
    def __init__(self, depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=-1, norm_eval=False,
                 style="pytorch", with_cp=False, norm_cfg=None, pretrained=None, init_cfg=None,
                 zero_init_residual=True, **kwargs):
        super().__init__(init_cfg)
        if depth not in _DEPTH or style != "pytorch":
            raise NotImplementedError(f"ResNet depth={depth} style={style}")
        self.depth, self.out_indices = depth, tuple(out_indices)
        self.frozen_stages, self.norm_eval, self.with_cp = frozen_stages, norm_eval, with_cp
        self.pretrained = pretrained
        self.zero_init_residual = zero_init_residual
        self.conv1 = Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        inplanes = 64
        self.res_layers = []
        for i, blocks in enumerate(_DEPTH[depth][:num_stages]):
            planes, stride = 64 * 2 ** i, 1 if i == 0 else 2
            down = None
            if stride != 1 or inplanes != planes * 4:
                down = nn.Sequential(Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                     BatchNorm2d(planes * 4))
            stage = [Bottleneck(inplanes, planes, stride, down, with_cp)]
            inplanes = planes * 4
            stage += [Bottleneck(inplanes, planes, with_cp=with_cp) for _ in range(1, blocks)]
            name = f"layer{i + 1}"
            self.add_module(name, nn.Sequential(*stage))
            self.res_layers.append(name)
        self._freeze()

    def _freeze(self):
        if self.frozen_stages >= 0:
            for m in (self.conv1, self.bn1):
                m.eval()
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f"layer{i}")
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def init_weights(self):
        """mmdet's ResNet initialisation: Kaiming-normal convolutions, unit BatchNorm, and (``zero_init_residual``,
        mmdet's default) the last BatchNorm of every bottleneck at zero; or the ``pretrained`` checkpoint.  A
        checkpoint path that is set but cannot be honoured raises -- a run must not silently start from random
        weights when the config asks for ImageNet ones (the files are loaded with ``weights_only=True``)."""
        if self.pretrained:
            load_checkpoint(self, self.pretrained, strict=False)
            return
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)

    def forward(self, x):
        x = self.maxpool(self.bn1(self.conv1(x), relu=True))
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        self._freeze()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self


class _ConvModule(nn.Module):
    """conv (+ BatchNorm) with mmcv.ConvModule's child names ``conv`` / ``bn``."""

    def __init__(self, cin, cout, k, padding=0, with_bn=False):
        super().__init__()
        self.conv = Conv2d(cin, cout, k, padding=padding, bias=not with_bn)
        if with_bn:
            self.bn = BatchNorm2d(cout)
        self.with_bn = with_bn

    def forward(self, x, out=None):
        x = self.conv(x)
        if out is not None:
            return self.bn(x, out=out)
        return self.bn(x) if self.with_bn else x


class FPN(BaseModule):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None, norm_cfg=None, act_cfg=None,
                 upsample_cfg=dict(mode="nearest"), init_cfg=None):
        super().__init__(init_cfg)
        self.in_channels, self.out_channels, self.num_outs = list(in_channels), out_channels, num_outs
        self.start_level = start_level
        self.backbone_end_level = len(in_channels) if end_level in (-1, len(in_channels) - 1) else end_level + 1
        used = self.backbone_end_level - start_level
        if num_outs != used:
            raise NotImplementedError("extra FPN levels are unused by the HiP-AD configs (num_outs == #inputs)")
        if act_cfg is not None or conv_cfg is not None:
            raise NotImplementedError("FPN act_cfg / conv_cfg")
        with_bn = norm_cfg is not None
        self.upsample_cfg = dict(upsample_cfg)
        self.lateral_convs = nn.ModuleList(
            _ConvModule(in_channels[i], out_channels, 1, with_bn=with_bn and not no_norm_on_lateral)
            for i in range(start_level, self.backbone_end_level))
        self.fpn_convs = nn.ModuleList(_ConvModule(out_channels, out_channels, 3, padding=1, with_bn=with_bn)
                                       for _ in range(used))

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, inputs, out_blocks=None):
        """``out_blocks`` (ours): a function (level index, conv output) -> (samples, rows, C) block of the flat pyramid,
        or None; when every output convolution ends in a norm layer on the fused kernels, that layer writes the level
        straight into its block and the level comes back as a (samples, cameras, C, h, w) view of the flat tensor."""
        lat = [conv(inputs[i + self.start_level]) for i, conv in enumerate(self.lateral_convs)]
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], **self.upsample_cfg)
        if out_blocks is None or not all(c.with_bn for c in self.fpn_convs):
            return tuple(conv(x) for conv, x in zip(self.fpn_convs, lat))
        ys = [conv.conv(x) for conv, x in zip(self.fpn_convs, lat)]
        if not all(conv.bn.fused_path(y) for conv, y in zip(self.fpn_convs, ys)):
            return tuple(conv.bn(y) for conv, y in zip(self.fpn_convs, ys))          # (N, C, h, w) levels, as without out_blocks
        return tuple(conv.bn(y, out=out_blocks(i, y)) for i, (conv, y) in enumerate(zip(self.fpn_convs, ys)))


if not HAVE_MMCV:
    BACKBONES.register_module("ResNet", module=ResNet)
    NECKS.register_module("FPN", module=FPN)
