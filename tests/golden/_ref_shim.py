"""Import shim for the reference's torch-only modules (BUILD CONTAINER ONLY).

Used solely by tests/golden/make_golden.py to run the reference's own Python code
(/root/reference, read-only, absent on the GPU box) and emit golden vectors.  Nothing
here is imported by tests at run time; nothing from the reference is copied: the shim
only provides stand-ins for third-party names (mmcv / mmdet / flash_attn) that are not
installed in this image, so that `importlib` can load the reference's modules from
where they lie.

Run with PYTHONDONTWRITEBYTECODE=1 so no __pycache__ is written into /root/reference.
"""
import functools
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REF = "/root/reference"
PLUGIN = os.path.join(REF, "projects", "mmdet3d_plugin")


class Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls

        if module is not None:
            return deco(module)
        return deco

    def get(self, key):
        return self.module_dict.get(key)

    def build(self, cfg, **kw):
        return build_from_cfg(cfg, self, kw or None)


_TORCH_TYPES = {"LN": nn.LayerNorm, "ReLU": nn.ReLU, "Dropout": nn.Dropout, "GELU": nn.GELU}


def build_from_cfg(cfg, registry, default_args=None):
    if cfg is None:
        return None
    cfg = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            cfg.setdefault(k, v)
    typ = cfg.pop("type")
    if isinstance(typ, str):
        cls = registry.get(typ) if registry is not None else None
        if cls is None:
            for reg in ALL_REGISTRIES:
                cls = reg.get(typ)
                if cls is not None:
                    break
        if cls is None and typ in _TORCH_TYPES:
            cls = _TORCH_TYPES[typ]
            if typ == "LN":
                return cls(cfg.pop("normalized_shape"), **cfg)
            if typ == "Dropout":
                return cls(cfg.pop("p", cfg.pop("drop_prob", 0.5)))
        if cls is None:
            raise KeyError(f"{typ} not registered")
    else:
        cls = typ
    return cls(**cfg)


ATTENTION = Registry("attention")
PLUGIN_LAYERS = Registry("plugin layer")
POSITIONAL_ENCODING = Registry("position encoding")
FEEDFORWARD_NETWORK = Registry("feed-forward Network")
NORM_LAYERS = Registry("norm layer")
HEADS = Registry("heads")
LOSSES = Registry("losses")
DETECTORS = Registry("detectors")
BBOX_SAMPLERS = Registry("bbox_sampler")
BBOX_CODERS = Registry("bbox_coder")
BBOX_ASSIGNERS = Registry("bbox_assigner")
MATCH_COST = Registry("match_cost")
ALL_REGISTRIES = [ATTENTION, PLUGIN_LAYERS, POSITIONAL_ENCODING, FEEDFORWARD_NETWORK, NORM_LAYERS,
                  HEADS, LOSSES, DETECTORS, BBOX_SAMPLERS, BBOX_CODERS, BBOX_ASSIGNERS, MATCH_COST]


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg

    def init_weights(self):
        pass


class Sequential(BaseModule, nn.Sequential):
    def __init__(self, *args, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.Sequential.__init__(self, *args)


class Scale(nn.Module):
    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        return x * self.scale


def bias_init_with_prob(p):
    import math
    return float(-math.log((1 - p) / p))


def xavier_init(module, gain=1, bias=0, distribution="normal"):
    if hasattr(module, "weight") and module.weight is not None:
        if distribution == "uniform":
            nn.init.xavier_uniform_(module.weight, gain=gain)
        else:
            nn.init.xavier_normal_(module.weight, gain=gain)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if hasattr(module, "weight") and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def build_activation_layer(cfg):
    cfg = dict(cfg)
    t = cfg.pop("type")
    return {"ReLU": nn.ReLU, "GELU": nn.GELU}[t](**cfg)


def build_norm_layer(cfg, num_features, postfix=""):
    cfg = dict(cfg)
    t = cfg.pop("type")
    assert t == "LN", t
    return "ln" + str(postfix), nn.LayerNorm(num_features, **cfg)


def build_dropout(cfg, default_args=None):
    cfg = dict(cfg)
    t = cfg.pop("type")
    assert t == "Dropout", t
    return nn.Dropout(cfg.pop("drop_prob", cfg.pop("p", 0.5)))


def _identity_decorator(*dargs, **dkw):
    if len(dargs) == 1 and callable(dargs[0]) and not dkw:
        return dargs[0]

    def deco(fn):
        return fn

    return deco


def deprecated_api_warning(name_dict, cls_name=None):
    def deco(fn):
        @functools.wraps(fn)
        def wrapper(*a, **k):
            return fn(*a, **k)
        return wrapper
    return deco


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _ns(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


_installed = False


def install():
    """Pre-seed sys.modules so the reference's leaf modules import without mmcv/mmdet."""
    global _installed
    if _installed:
        return
    _installed = True
    sys.dont_write_bytecode = True
    # namespace stubs: package __init__ files (which pull datasets/apis) never run
    _ns("projects", os.path.join(REF, "projects"))
    _ns("projects.mmdet3d_plugin", PLUGIN)
    _ns("projects.mmdet3d_plugin.core", os.path.join(PLUGIN, "core"))
    _ns("projects.mmdet3d_plugin.models", os.path.join(PLUGIN, "models"))
    for sub in ("det", "map", "plan", "ego", "motion"):
        _ns(f"projects.mmdet3d_plugin.models.{sub}", os.path.join(PLUGIN, "models", sub))
    _ns("projects.mmdet3d_plugin.datasets", os.path.join(PLUGIN, "datasets"))
    _ns("projects.mmdet3d_plugin.datasets.pipelines", os.path.join(PLUGIN, "datasets", "pipelines"))
    # ops/__init__.py itself is torch-only; only its two compiled extension leaves are stubbed
    _mod("projects.mmdet3d_plugin.ops.deformable_aggregation_ext")
    _mod("projects.mmdet3d_plugin.ops.deformable_aggregation_a800_ext")
    _mod("projects.mmdet3d_plugin.datasets.pipelines.vectorize_numpy", VectorizeMapNumpy=object)

    Linear = nn.Linear
    _mod("mmcv")
    _mod("mmcv.cnn", Linear=Linear, Scale=Scale, bias_init_with_prob=bias_init_with_prob,
         build_activation_layer=build_activation_layer, build_norm_layer=build_norm_layer,
         xavier_init=xavier_init, constant_init=constant_init)
    _mod("mmcv.cnn.bricks")
    _mod("mmcv.cnn.bricks.registry", ATTENTION=ATTENTION, PLUGIN_LAYERS=PLUGIN_LAYERS,
         POSITIONAL_ENCODING=POSITIONAL_ENCODING, FEEDFORWARD_NETWORK=FEEDFORWARD_NETWORK,
         NORM_LAYERS=NORM_LAYERS)
    _mod("mmcv.cnn.bricks.drop", build_dropout=build_dropout)
    _mod("mmcv.cnn.bricks.transformer", FFN=object,
         build_attention=lambda cfg: build_from_cfg(cfg, ATTENTION),
         build_feedforward_network=lambda cfg: build_from_cfg(cfg, FEEDFORWARD_NETWORK))
    _mod("mmcv.utils", build_from_cfg=build_from_cfg, deprecated_api_warning=deprecated_api_warning,
         Registry=Registry)
    _mod("mmcv.runner", BaseModule=BaseModule, force_fp32=_identity_decorator, auto_fp16=_identity_decorator,
         Sequential=Sequential)
    _mod("mmcv.runner.base_module", BaseModule=BaseModule, Sequential=Sequential)
    _mod("mmdet")
    _mod("mmdet.core.bbox")
    _mod("mmdet.core.bbox.builder", BBOX_SAMPLERS=BBOX_SAMPLERS, BBOX_CODERS=BBOX_CODERS,
         BBOX_ASSIGNERS=BBOX_ASSIGNERS)
    import _mmdet_losses as ML  # restated mmdet==2.28.2 loss primitives (third-party, absent here)
    for cls in (ML.FocalLoss, ML.L1Loss, ML.CrossEntropyLoss, ML.GaussianFocalLoss):
        LOSSES.register_module(module=cls)
    MATCH_COST.register_module(module=ML.FocalLossCost)
    build_match_cost = lambda cfg: build_from_cfg(cfg, MATCH_COST)  # noqa: E731
    build_assigner = lambda cfg, **kw: build_from_cfg(cfg, BBOX_ASSIGNERS)  # noqa: E731
    _mod("mmdet.core", reduce_mean=lambda x: x, build_assigner=build_assigner,
         build_sampler=lambda cfg, **kw: build_from_cfg(cfg, BBOX_SAMPLERS))
    _mod("mmdet.core.bbox.assigners", AssignResult=ML.AssignResult, BaseAssigner=ML.BaseAssigner)
    _mod("mmdet.core.bbox.match_costs", build_match_cost=build_match_cost)
    _mod("mmdet.core.bbox.match_costs.builder", MATCH_COST=MATCH_COST)
    _mod("mmdet.models", HEADS=HEADS, LOSSES=LOSSES, DETECTORS=DETECTORS,
         build_loss=lambda cfg: build_from_cfg(cfg, LOSSES))
    _mod("mmdet.models.builder", LOSSES=LOSSES, HEADS=HEADS, DETECTORS=DETECTORS)
    _mod("mmdet.models.losses", l1_loss=ML.l1_loss, smooth_l1_loss=ML.smooth_l1_loss)
    _mod("cv2")  # datasets/utils.py imports it for drawing helpers the decoders never call
    _mod("flash_attn")
    _mod("flash_attn.flash_attn_interface", flash_attn_unpadded_kvpacked_func=None,
         flash_attn_varlen_kvpacked_func=None)
    _mod("flash_attn.bert_padding", unpad_input=None, pad_input=None, index_first_axis=None)


def ref_import(name):
    """import_module('projects.mmdet3d_plugin.' + name) from the reference tree."""
    install()
    return importlib.import_module("projects.mmdet3d_plugin." + name)
