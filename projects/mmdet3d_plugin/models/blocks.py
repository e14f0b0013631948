"""Decoder building blocks on the MI355X path.

Same registered names, constructor keywords and parameter names as the reference's
``projects/mmdet3d_plugin/models/blocks.py`` (so its configs build these classes unchanged and
its checkpoints load), different insides:

* ``DeformableFeatureAggregation`` (reference blocks.py:45-264) runs three HIP kernels --
  projection straight into the op layout, fused softmax-weights, the aggregation op -- and
  applies ``weights_fc`` to the anchor part and the camera part separately
  (W(f_a + c_k) + b = (W f_a + b) + W c_k: A + cams GEMM rows instead of A * cams).
* ``AsymmetricFFN`` (reference blocks.py:328-396) and ``DenseDepthNet`` (blocks.py:267-325).

There is no grid_sample fallback here: ``use_deformable_func=False`` (the reference's CPU path,
blocks.py:162-170) raises -- the oracle of that path lives in tests/golden + oracle/.
"""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F

from hipad_amd import functional as HF
from hipad_amd.compat import (MLPStack, ATTENTION, FEEDFORWARD_NETWORK, PLUGIN_LAYERS, BaseModule, FusedReLU, LayerNorm, Linear, Sequential,
                              linear_relu,
                              build_activation_layer, build_dropout, build_from_cfg, build_norm_layer,
                              constant_init, xavier_init)

from ..ops import deformable_aggregation_function as DAF

__all__ = ["DeformableFeatureAggregation", "DenseDepthNet", "AsymmetricFFN", "linear_relu_ln", "CustomOperation",
           "score_head", "mlp_head"]


def linear_relu_ln(embed_dims, in_loops, out_loops, input_dims=None):
    """[ (Linear, ReLU) x in_loops, LayerNorm ] x out_loops as a flat list (reference blocks.py:32-42)."""
    width_in = embed_dims if input_dims is None else input_dims
    stack = []
    for _ in range(out_loops):
        for _ in range(in_loops):
            stack += list(linear_relu(width_in, embed_dims))  # ReLU fused into the GEMM epilogue
            width_in = embed_dims
        stack.append(LayerNorm(embed_dims))
    return stack


def score_head(embed_dims, out_dim=1):
    """[Linear, ReLU, LayerNorm] x 2 then a Linear to ``out_dim`` logits (Sequential indices 0..6)."""
    return MLPStack(*linear_relu_ln(embed_dims, 1, 2), Linear(embed_dims, out_dim))


def mlp_head(embed_dims, out_dim):
    """[Linear, ReLU] x 2 then a Linear to ``out_dim`` values (Sequential indices 0..4)."""
    return MLPStack(*linear_relu(embed_dims, embed_dims), *linear_relu(embed_dims, embed_dims), Linear(embed_dims, out_dim))


@ATTENTION.register_module()
class DeformableFeatureAggregation(BaseModule):
    def __init__(self, embed_dims: int = 256, num_groups: int = 8, num_levels: int = 4, num_sample: int = 20,
                 num_cams: int = 6, proj_drop: float = 0.0, attn_drop: float = 0.0, kps_generator: dict = None,
                 temporal_fusion_module=None, use_temporal_anchor_embed=True, use_deformable_func=False,
                 use_camera_embed=False, use_points_embed=False, use_anchor_embed=False, residual_mode="add"):
        super().__init__()
        if embed_dims % num_groups:
            raise ValueError(f"embed_dims must be divisible by num_groups, but got {embed_dims} and {num_groups}")
        if use_points_embed or use_anchor_embed:
            raise NotImplementedError("point-embedding variants of the aggregation module are not on the hot path "
                                      "(off in projects/configs/hipad_b2d_stage{1,2}.py)")
        if temporal_fusion_module is not None:
            raise NotImplementedError("temporal_fusion_module is unused by the HiP-AD configs")
        self.embed_dims, self.num_groups, self.num_levels = embed_dims, num_groups, num_levels
        self.num_cams, self.num_sample = num_cams, num_sample
        self.group_dims = embed_dims // num_groups
        self.attn_drop, self.residual_mode = attn_drop, residual_mode
        self.use_deformable_func = use_deformable_func
        self.use_camera_embed, self.use_points_embed = use_camera_embed, use_points_embed
        self.use_temporal_anchor_embed = use_temporal_anchor_embed
        self.proj_drop = nn.Dropout(proj_drop)

        kps_cfg = dict(kps_generator)
        kps_cfg["embed_dims"] = embed_dims
        self.kps_generator = build_from_cfg(kps_cfg, PLUGIN_LAYERS)
        self.kps_pts = self.num_pts = self.kps_generator.num_pts
        self.temp_module = None
        self.output_proj = Linear(embed_dims, embed_dims)
        per_cam = num_groups * num_levels * self.num_pts
        if use_camera_embed:
            self.camera_encoder = MLPStack(*linear_relu_ln(embed_dims, 1, 2, 12))
            self.weights_fc = Linear(embed_dims, per_cam)
        else:
            self.camera_encoder = None
            self.weights_fc = Linear(embed_dims, per_cam * num_cams)

    def init_weight(self):
        constant_init(self.weights_fc, val=0.0, bias=0.0)
        xavier_init(self.output_proj, distribution="uniform", bias=0.0)

    # -- sampling weights: (bs, A, P, cams, L, G), the aggregation op's layout ---------------
    def _keep_mask(self, bs, num_anchor, device):
        if not (self.training and self.attn_drop > 0):
            return None
        # one Bernoulli draw per (anchor, camera, point), shared by levels and groups, scaled to
        # keep the expectation (reference blocks.py:209-212; drawn on the device here)
        if torch.device(device).type == "cuda":
            from hipad_amd import lib as _lib
            if not hasattr(self, "_mask_seed"):
                self._mask_seed = HF.new_call_site_seed()
            # one launch, a fresh draw per step from the device-side dropout clock (replayable from a hipGraph)
            return _lib.keep_mask((bs, num_anchor, self.num_cams, self.num_pts), self.attn_drop, self._mask_seed,
                                  HF.dropout_clock(device), device)
        keep = torch.rand(bs, num_anchor, self.num_cams, self.num_pts, device=device) > self.attn_drop
        return keep.float() / (1.0 - self.attn_drop)

    def _get_weights(self, instance_feature, anchor_embed, metas=None, op_layout=False):
        """Reference-layout weights (bs, A, cams, L, P, G) (blocks.py:178-214); ``op_layout=True``
        returns the permuted (bs, A, P, cams, L, G) tensor the kernels use (no extra copy)."""
        bs, num_anchor = instance_feature.shape[:2]
        feature = instance_feature + anchor_embed
        L, P, G = self.num_levels, self.num_pts, self.num_groups
        keep = self._keep_mask(bs, num_anchor, feature.device)
        if self.camera_encoder is not None:
            cam_in = metas["projection_mat"][:, :, :3].reshape(bs, self.num_cams, -1)
            cam_embed = getattr(self, "_cam_embed", None)     # precomputed for the whole frame by the decoder (one launch
            if cam_embed is None:                              # for all modules' camera encoders), else computed here
                cam_embed = self.camera_encoder(cam_in.to(feature.dtype))
            u = self.weights_fc(feature)                       # (bs, A, n): anchor part + bias
            v = HF.linear(cam_embed, self.weights_fc.weight)   # (bs, cams, n): camera part
            w = HF.sampling_weights(u, v, keep, L, P, G)
        else:
            u = self.weights_fc(feature).reshape(bs, num_anchor, self.num_cams, L * P * G)
            w = HF.sampling_weights(u, None, keep, L, P, G)
        return w if op_layout else w.permute(0, 1, 3, 4, 2, 5)

    @staticmethod
    def project_points(key_points, projection_mat, image_wh=None):
        """(bs, cams, A, P, 2) like the reference's static method (blocks.py:216-225)."""
        return HF.project_points(key_points, projection_mat, image_wh).permute(0, 3, 1, 2, 4)

    def forward(self, instance_feature: torch.Tensor, anchor: torch.Tensor, anchor_embed: torch.Tensor,
                feature_maps: List[torch.Tensor], metas: dict, **kwargs):
        if not self.use_deformable_func:
            raise RuntimeError("DeformableFeatureAggregation(use_deformable_func=False) selects the reference's "
                               "grid_sample CPU path; this build has only the HIP op (set use_deformable_func=True)")
        bs, num_anchor = instance_feature.shape[:2]
        # NB the reference passes (anchor, anchor_embed, instance_feature) positionally; the box
        # generator's second parameter is named instance_feature, so its learnable offsets are a
        # function of anchor_embed -- kept as is.
        weights = self._get_weights(instance_feature, anchor_embed, metas, op_layout=True)
        fused = anchor.is_cuda and hasattr(self.kps_generator, "project") and not getattr(self.kps_generator, "with_points_embed", False)
        if fused:
            # generator + projection fused: the key points are never written to memory.  (Positional call like the
            # reference: the box generator's second parameter receives anchor_embed, the poly-line generator's
            # second and third receive anchor_embed and instance_feature.)
            extra = (anchor_embed,) if self.kps_generator.project.__code__.co_argcount == 5 else (anchor_embed, instance_feature)
            loc = self.kps_generator.project(anchor, *extra, metas["projection_mat"], metas.get("image_wh"))
        else:
            key_points = self.kps_generator(anchor, anchor_embed, instance_feature)
            loc = HF.project_points(key_points, metas["projection_mat"], metas.get("image_wh"))
        features = DAF(*feature_maps, loc, weights).reshape(bs, num_anchor, self.embed_dims)
        output = self.proj_drop(self.output_proj(features))
        if self.residual_mode == "add":
            output = output + instance_feature
        elif self.residual_mode == "cat":
            output = torch.cat([output, instance_feature], dim=-1)
        return output


@PLUGIN_LAYERS.register_module()
class DenseDepthNet(BaseModule):
    """Auxiliary dense-depth heads: one 1x1 conv per pyramid level, exp, focal scaling."""

    def __init__(self, embed_dims=256, num_depth_layers=1, equal_focal=100, max_depth=60, loss_weight=1.0):
        super().__init__()
        self.embed_dims, self.num_depth_layers = embed_dims, num_depth_layers
        self.equal_focal, self.max_depth, self.loss_weight = equal_focal, max_depth, loss_weight
        self.depth_layers = nn.ModuleList(nn.Conv2d(embed_dims, 1, kernel_size=1) for _ in range(num_depth_layers))

    def forward(self, feature_maps, focal=None, gt_depths=None):
        scale = None if focal is None else focal.reshape(-1, 1, 1, 1) / self.equal_focal
        depths = []
        for head, feat in zip(self.depth_layers, feature_maps[: self.num_depth_layers]):
            d = head(feat.flatten(end_dim=1).float()).exp()
            depths.append(d if scale is None else d * scale)
        if gt_depths is not None and self.training:
            return self.loss(depths, gt_depths)
        return depths

    def on_pyramid(self, flat, blocks, num_cams, levels, focal=None):
        """The training-time handle of the fused path (hipad_amd/csrc/depthloss.hip): nothing is computed until
        ``loss(handle, gt_depths)``, which reads the rows of the frame's flat bf16 pyramid ``flat`` (bs, rows, 256) in
        place -- ``blocks[l]`` = (first row, rows per sample) of level ``l`` -- and adds the feature gradient into the
        frame's shared pyramid-gradient buffer.  Iterating the handle evaluates the heads the module way."""
        return PyramidDepth(self, flat, blocks, num_cams, levels, focal)

    def loss(self, depth_preds, gt_depths):
        if isinstance(depth_preds, PyramidDepth):
            h = depth_preds
            n = min(self.num_depth_layers, len(gt_depths))
            geometry = [(h.blocks[l][1] // h.num_cams, h.blocks[l][0]) for l in range(n)]
            return HF.depth_loss(h.flat, h.focal, list(gt_depths[:n]), [m.weight for m in self.depth_layers[:n]],
                                 [m.bias for m in self.depth_layers[:n]], geometry, h.num_cams, self.equal_focal,
                                 self.max_depth, self.loss_weight)
        total = 0.0
        for pred, gt in zip(depth_preds, gt_depths):
            pred = pred.permute(0, 2, 3, 1).reshape(-1)
            gt = gt.reshape(-1)
            valid = (gt > 0.0) & ~torch.isnan(pred)
            # masked sum instead of the reference's boolean indexing (blocks.py:313-316): same value, no
            # data-dependent shape and no host synchronisation (the step stays capturable)
            diff = (pred.float().clamp(0.0, self.max_depth) - gt.float()).abs()
            err = torch.where(valid, diff, torch.zeros_like(diff)).sum()
            count = torch.clamp(valid.sum().float() * len(depth_preds), min=1.0)
            total = total + err / count * self.loss_weight
        return total


class PyramidDepth:
    """What DenseDepthNet.on_pyramid returns (see there)."""

    def __init__(self, net, flat, blocks, num_cams, levels, focal):
        self.net, self.flat, self.blocks, self.num_cams, self.levels, self.focal = net, flat, blocks, num_cams, levels, focal

    def __iter__(self):
        return iter(self.net(self.levels, self.focal))


@FEEDFORWARD_NETWORK.register_module()
class AsymmetricFFN(BaseModule):
    """pre-norm -> Linear/act/drop x (num_fcs-1) -> Linear -> drop, plus an identity branch that is
    itself a Linear when input and output widths differ (reference blocks.py:328-396)."""

    def __init__(self, in_channels=None, pre_norm=None, embed_dims=256, feedforward_channels=1024, num_fcs=2,
                 act_cfg=dict(type="ReLU", inplace=True), ffn_drop=0.0, dropout_layer=None, add_identity=True,
                 init_cfg=None, **kwargs):
        super().__init__(init_cfg)
        if num_fcs < 2:
            raise ValueError(f"num_fcs should be no less than 2. got {num_fcs}.")
        self.in_channels, self.embed_dims = in_channels, embed_dims
        self.feedforward_channels, self.num_fcs, self.act_cfg = feedforward_channels, num_fcs, act_cfg
        self.activate = build_activation_layer(act_cfg)
        width = embed_dims if in_channels is None else in_channels
        self.pre_norm = build_norm_layer(pre_norm, width)[1] if pre_norm is not None else None
        stages = []
        relu_act = isinstance(self.activate, nn.ReLU)
        for _ in range(num_fcs - 1):
            if relu_act:
                first, act = linear_relu(width, feedforward_channels)
            else:
                first, act = Linear(width, feedforward_channels), self.activate
            stages.append(Sequential(first, act, nn.Dropout(ffn_drop)))
            width = feedforward_channels
        stages += [Linear(feedforward_channels, embed_dims), nn.Dropout(ffn_drop)]
        self.layers = Sequential(*stages)
        self.dropout_layer = build_dropout(dropout_layer) if dropout_layer else nn.Identity()
        self.add_identity = add_identity
        if add_identity:
            # the reference compares the (by now overwritten) hidden width with embed_dims, so with
            # in_channels given the identity branch is always a Linear(in_channels, embed_dims)
            self.identity_fc = nn.Identity() if width == embed_dims else Linear(self.in_channels, embed_dims)

    def forward(self, x, identity=None):
        if self.pre_norm is not None:
            x = self.pre_norm(x)
        out = self.layers(x)
        if not self.add_identity:
            return self.dropout_layer(out)
        base = self.identity_fc(x if identity is None else identity)
        if isinstance(self.dropout_layer, nn.Dropout) and out.is_cuda:
            if not hasattr(self, "_drop_seed"):
                self._drop_seed = HF.new_call_site_seed()
            return HF.dropout_add(out, base, self.dropout_layer.p, self._drop_seed, self.training)   # one launch
        return base + self.dropout_layer(out)


@PLUGIN_LAYERS.register_module()
class CustomOperation(BaseModule):
    """Parameter-free placeholder the decoder program uses for concat / split / deformable / refine slots."""

    def __init__(self):
        super().__init__()
        self.identity = nn.Identity()

    def forward(self, feature):
        return self.identity(feature)
