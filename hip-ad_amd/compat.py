"""Minimal stand-in for the parts of mmcv / mmdet the hot path's modules are built with.

The reference instantiates every module by name through mmcv registries from Python-dict
configs (``build_from_cfg(cfg, REGISTRY)``, projects/mmdet3d_plugin/models/sparse_onedecoder.py:203-206)
and subclasses ``mmcv.runner.BaseModule``.  mmcv / mmdet are third-party packages that are not
installed here (and the GPU box receives only this repo), so this file provides the handful of
names the path needs, with the same call signatures.  When the real packages are importable
they are used instead, so the plugin registers into the real registries under tools/train.py.
"""
import importlib.util
import math
import os
import types

import torch
import torch.nn as nn

try:  # prefer the real thing when present (drop-in under the reference's tools/)
    from mmcv.utils import Registry, build_from_cfg  # type: ignore
    from mmcv.cnn.bricks.registry import (ATTENTION, FEEDFORWARD_NETWORK, NORM_LAYERS,  # type: ignore
                                          PLUGIN_LAYERS, POSITIONAL_ENCODING)
    from mmcv.runner.base_module import BaseModule, Sequential  # type: ignore
    from mmdet.models import DETECTORS, HEADS, LOSSES  # type: ignore
    from mmdet.core.bbox.builder import BBOX_CODERS, BBOX_SAMPLERS  # type: ignore
    HAVE_MMCV = True
except Exception:  # noqa: BLE001 - any import problem means "not available"
    HAVE_MMCV = False

class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm (same parameters / state_dict) on the HIP kernel for CUDA fp32 inputs: one forward and ONE
    backward launch, gamma / beta gradients accumulated in place (hipad_amd.functional.layer_norm)."""

    def forward(self, x):
        from . import functional as HF
        if x.is_cuda and len(self.normalized_shape) == 1 and self.weight is not None:
            return HF.layer_norm(x, self.weight, self.bias, self.eps)
        if x.is_cuda:
            HF._library("LayerNorm module (normalized_shape %s, affine %s)" % (tuple(self.normalized_shape), self.weight is not None))
        return super().forward(x)


if not HAVE_MMCV:

    class Registry:
        """name -> class table with mmcv's ``register_module`` decorator protocol."""

        def __init__(self, name):
            self._name = name
            self._table = {}

        @property
        def name(self):
            return self._name

        @property
        def module_dict(self):
            return self._table

        def get(self, key):
            return self._table.get(key)

        def __contains__(self, key):
            return key in self._table

        def _add(self, cls, name=None, force=False):
            key = name or cls.__name__
            if not force and key in self._table and self._table[key] is not cls:
                raise KeyError(f"{key} is already registered in {self._name}")
            self._table[key] = cls

        def register_module(self, name=None, force=False, module=None):
            if module is not None:
                self._add(module, name, force)
                return module

            def decorate(cls):
                self._add(cls, name, force)
                return cls

            return decorate

        def build(self, cfg, default_args=None):
            return build_from_cfg(cfg, self, default_args)

    _ALL = []

    def _registry(name):
        r = Registry(name)
        _ALL.append(r)
        return r

    ATTENTION = _registry("attention")
    PLUGIN_LAYERS = _registry("plugin layer")
    POSITIONAL_ENCODING = _registry("position encoding")
    FEEDFORWARD_NETWORK = _registry("feed-forward network")
    NORM_LAYERS = _registry("norm layer")
    HEADS = _registry("head")
    DETECTORS = _registry("detector")
    LOSSES = _registry("loss")
    BBOX_SAMPLERS = _registry("bbox sampler")
    BBOX_CODERS = _registry("bbox coder")
    BACKBONES = _registry("backbone")
    NECKS = _registry("neck")
    PIPELINES = _registry("pipeline")

    NORM_LAYERS.register_module("LN", module=LayerNorm)

    def build_from_cfg(cfg, registry, default_args=None):
        """Instantiate ``cfg['type']`` (a registered name or a class) with the remaining keys."""
        if cfg is None:
            return None
        if not isinstance(cfg, dict) or "type" not in cfg:
            raise TypeError(f"cfg must be a dict with a 'type' key, got {cfg!r}")
        args = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        kind = args.pop("type")
        if isinstance(kind, str):
            cls = registry.get(kind) if registry is not None else None
            if cls is None:  # mmcv registries have parent/child scopes; ours are flat: search all
                for r in _ALL:
                    cls = r.get(kind)
                    if cls is not None:
                        break
            if cls is None:
                raise KeyError(f"{kind} is not registered (looked in {getattr(registry, 'name', None)} and all others)")
        else:
            cls = kind
        return cls(**args)

    class BaseModule(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()
            self.init_cfg = init_cfg

        def init_weights(self):
            for m in self.children():
                if hasattr(m, "init_weights"):
                    m.init_weights()

    class Sequential(BaseModule, nn.Sequential):
        def __init__(self, *mods, init_cfg=None):
            BaseModule.__init__(self, init_cfg)
            nn.Sequential.__init__(self, *mods)
else:
    from mmdet.models import BACKBONES, NECKS  # type: ignore  # noqa: F401
    from mmdet.datasets.builder import PIPELINES  # type: ignore  # noqa: F401


class Linear(nn.Linear):
    """nn.Linear (same parameters / state_dict) that runs on the hand-written MFMA kernel for CUDA
    inputs when ``hipad_amd.functional.LINEAR_MODE == "mfma_bf16"`` (bias and an optional ReLU fused in
    the epilogue, weight/bias gradients accumulated in place), and on torch's fp32 GEMM otherwise."""

    fuse_relu = False

    def forward(self, x):
        from . import functional as HF
        if x.is_cuda and HF.LINEAR_MODE == "mfma_bf16":
            return HF.linear(x, self.weight, self.bias, relu=self.fuse_relu)
        if x.is_cuda:
            HF._library("Linear module (mode %s)" % HF.LINEAR_MODE)
        y = nn.functional.linear(x, self.weight, self.bias)
        return nn.functional.relu(y) if self.fuse_relu else y


class MLPStack(nn.Sequential):
    """nn.Sequential (same children, same state_dict keys) that can run every [Linear(+ReLU), FusedReLU, LayerNorm]
    triple as ONE forward launch (hipad_linear_relu_ln_forward) -- opt-in, see functional.FUSE_LINEAR_LN."""

    def forward(self, x, x1=None, residual=None):
        """``x1`` (same shape as x) is added to the input, ``residual`` (shape of the output) to the result -- both inside
        the chain kernel on the GPU path (they replace the separate add kernels around the stack)."""
        from . import chain as CH
        from . import functional as HF
        if CH.usable(x):
            spec = CH.spec_of(self)
            if spec is not None:
                return CH.run([CH.Call(spec, x, x1, residual)])[0]
        if x1 is not None:
            x = x + x1
        y = self._forward_layers(x)
        return y if residual is None else y + residual

    def _forward_layers(self, x):
        from . import functional as HF
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if (i + 2 < len(mods) and isinstance(m, Linear) and m.fuse_relu and isinstance(mods[i + 1], FusedReLU)
                    and isinstance(mods[i + 2], LayerNorm) and HF.FUSE_LINEAR_LN
                    and HF.linear_relu_ln_ok(x, m.weight, mods[i + 2].weight)):
                ln = mods[i + 2]
                x = HF.linear_relu_ln(x, m.weight, m.bias, ln.weight, ln.bias, ln.eps)
                i += 3
            else:
                x = m(x)
                i += 1
        return x


def linear_relu(in_features, out_features):
    """(Linear with the ReLU fused into its epilogue, placeholder) -- two modules, so Sequential
    indices (and therefore state_dict keys) match a plain [Linear, ReLU] pair."""
    lin = Linear(in_features, out_features)
    lin.fuse_relu = True
    return lin, FusedReLU()


class FusedReLU(nn.Module):
    """Occupies the ReLU slot after a Linear whose kernel already applied the ReLU."""

    def forward(self, x):
        return x


class Scale(nn.Module):
    """Learnable per-channel (or scalar) multiplier (mmcv.cnn.Scale)."""

    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        return x * self.scale


def bias_init_with_prob(prior_prob):
    return float(-math.log((1 - prior_prob) / prior_prob))


def xavier_init(module, gain=1, bias=0, distribution="normal"):
    if getattr(module, "weight", None) is not None:
        (nn.init.xavier_uniform_ if distribution == "uniform" else nn.init.xavier_normal_)(module.weight, gain=gain)
    if getattr(module, "bias", None) is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if getattr(module, "weight", None) is not None:
        nn.init.constant_(module.weight, val)
    if getattr(module, "bias", None) is not None:
        nn.init.constant_(module.bias, bias)


def build_activation_layer(cfg):
    args = dict(cfg)
    kind = args.pop("type")
    table = {"ReLU": nn.ReLU, "GELU": nn.GELU, "Sigmoid": nn.Sigmoid, "Tanh": nn.Tanh}
    return table[kind](**args)


def build_norm_layer(cfg, num_features, postfix=""):
    args = dict(cfg)
    kind = args.pop("type")
    if kind == "LN":
        return f"ln{postfix}", LayerNorm(num_features, **args)
    if kind == "BN":
        args.pop("requires_grad", None)
        return f"bn{postfix}", nn.BatchNorm2d(num_features, **args)
    raise KeyError(kind)


def build_dropout(cfg, default_args=None):
    args = dict(cfg)
    kind = args.pop("type")
    if kind != "Dropout":
        raise KeyError(kind)
    return nn.Dropout(args.pop("drop_prob", args.pop("p", 0.5)))


def force_fp32(*dargs, **dkw):
    """Decorator kept for signature compatibility; our modules manage precision explicitly."""
    if len(dargs) == 1 and callable(dargs[0]) and not dkw:
        return dargs[0]
    return lambda fn: fn


auto_fp16 = force_fp32


def _identity_choice(tag, choice):
    return choice


# Every data-dependent DISCRETE choice of the path (temporal top-k selections, the motion-mode anchors' class, Hungarian
# indices, point order of a matched poly-line, winning trajectory mode, class-score gate on regression positives) passes
# through ``discrete_choice[0](tag, index_tensor)``.  Identity in the product; parity tests swap in a recorder / replayer
# so that two numerically different runs are compared on the same choices instead of on near-ties that fell the other way.
discrete_choice = [_identity_choice]
# "all": the hook sees every tag, and code paths that make such choices INSIDE a fused kernel (the fused training objective:
# assignment, line order, winning mode, class gate) step aside for the torch-op formulation while a hook is installed.
# "decoder": the hook only cares about the decoder's own choices ("topk", "motion_class"); the fused objective stays on.
discrete_scope = ["all"]


def discrete(tag, choice):
    return discrete_choice[0](tag, choice)


class CountExchange:
    """Cross-rank mean of the losses' positive counts (mmdet ``reduce_mean``, reference sparse_onedecoder.py:1134, 1190,
    1292) for a step whose kernels are captured in hipGraphs: a collective cannot sit inside a captured graph, so the
    counts go through a static buffer in three phases:

      collect  ``reduce_mean`` copies each local count vector into the buffer (inside graph 1: the forward + the target
               assignment) and returns the local value;
      exchange ONE eager all-reduce of the buffer between the graphs (``all_reduce``), mean over ranks;
      use      ``reduce_mean`` hands out the reduced values from the buffer, in the same call order (inside graph 2: the
               loss arithmetic + backward).

    Eager steps run the same three phases, so eager and replayed steps normalise identically.  ``mode`` is "direct"
    outside such a step: the collective runs at the call (and raises under capture instead of silently skipping)."""

    SIZE = 256

    def __init__(self):
        self.mode, self.buf, self.cursor, self.filled = "direct", None, 0, 0

    def _buffer(self, device):
        if self.buf is None or self.buf.device != device:
            self.buf = torch.zeros(self.SIZE, dtype=torch.float32, device=device)
        return self.buf

    def begin(self, mode):
        if mode not in ("direct", "collect", "use"):
            raise ValueError(mode)
        if mode == "use" and self.filled == 0:
            raise RuntimeError("CountExchange: 'use' phase without a preceding 'collect' phase")
        if mode == "collect":
            self.filled = 0
        self.mode, self.cursor = mode, 0

    def all_reduce(self, group=None):
        """Mean over the ranks of everything collected (one collective; a no-op for a single process)."""
        import torch.distributed as dist
        if self.filled and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            part = self.buf[: self.filled]
            dist.all_reduce(part, group=group)
            part.div_(dist.get_world_size(group))

    def exchange(self, tensor):
        n = tensor.numel()
        buf = self._buffer(tensor.device)
        if self.cursor + n > buf.numel():
            raise RuntimeError("CountExchange: buffer too small")
        part = buf[self.cursor:self.cursor + n]
        self.cursor += n
        if self.mode == "collect":
            part.copy_(tensor.detach().reshape(-1).to(torch.float32))
            self.filled = max(self.filled, self.cursor)
            return tensor
        return part.view(tensor.shape).to(tensor.dtype)


count_exchange = CountExchange()


def reduce_mean(tensor):
    """All-reduce(mean) across ranks when torch.distributed is initialised (mmdet.core.reduce_mean); inside a
    CountExchange step the value travels through its static buffer (see there)."""
    import torch.distributed as dist
    if count_exchange.mode != "direct":
        return count_exchange.exchange(tensor)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tensor
    if tensor.is_cuda and torch.cuda.is_current_stream_capturing():
        raise RuntimeError("reduce_mean: a collective cannot run inside a captured hipGraph; run the step through "
                           "hipad_amd.compat.count_exchange (collect / all_reduce / use), as hipad_amd.frame does")
    tensor = tensor.clone()
    dist.all_reduce(tensor.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
    return tensor


# ------------------------------------------------------------------------------------------
# Python-file configs (mmcv.Config.fromfile for the subset the reference's configs use)
# ------------------------------------------------------------------------------------------
class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: _wrap(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [_wrap(v) for v in obj]
    if isinstance(obj, tuple):
        return tuple(_wrap(v) for v in obj)
    return obj


class Config:
    """``Config.fromfile(path, overrides=...)``: executes a Python config file and exposes its
    public names; ``replace`` substitutes source text first (the reference configs hard-code an
    absolute ``project_dir``, projects/configs/hipad_b2d_stage2.py:77)."""

    def __init__(self, cfg_dict, filename=None):
        object.__setattr__(self, "_cfg", _wrap(cfg_dict))
        object.__setattr__(self, "filename", filename)

    @staticmethod
    def fromfile(path, replace=None, overrides=None):
        with open(path) as f:
            src = f.read()
        for old, new in (replace or {}).items():
            src = src.replace(old, new)
        ns = {"__file__": os.path.abspath(path)}
        exec(compile(src, path, "exec"), ns)
        cfg = {k: v for k, v in ns.items() if not k.startswith("_") and not isinstance(v, types.ModuleType)
               and not callable(v)}
        cfg = Config(cfg, path)
        for dotted, val in (overrides or {}).items():
            cfg.set(dotted, val)
        return cfg

    def set(self, dotted, val):
        node = self._cfg
        keys = dotted.split(".")
        for k in keys[:-1]:
            node = node[k]
        node[keys[-1]] = val

    def __getattr__(self, k):
        return getattr(self._cfg, k)

    def __getitem__(self, k):
        return self._cfg[k]

    def get(self, k, default=None):
        return self._cfg.get(k, default)
