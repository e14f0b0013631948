"""Poly-line (map / plan way-point) query blocks: anchor encoder, refinement head, key-point generator.

Registered names / keywords / parameter names follow the reference's
``projects/mmdet3d_plugin/models/map/blocks.py`` (encoder :18-42, refinement :80-135, key points
:137-225); the code is written for this repo.
"""
from typing import Tuple

import numpy as np
import torch
import torch.nn as nn

from hipad_amd.compat import (MLPStack, PLUGIN_LAYERS, POSITIONAL_ENCODING, BaseModule, Linear, Scale, bias_init_with_prob,
                              xavier_init)

from ..blocks import linear_relu_ln

__all__ = ["SparsePoint3DRefinementModule", "SparsePoint3DKeyPointsGenerator", "SparsePoint3DEncoder",
           "KeyPoint3DEncoder"]


@POSITIONAL_ENCODING.register_module()
class SparsePoint3DEncoder(BaseModule):
    """MLP over the flattened poly-line coordinates (reference map/blocks.py:18-42)."""

    def __init__(self, embed_dims: int = 256, num_sample: int = 20, coords_dim: int = 2,
                 return_points_embed: bool = False):
        super().__init__()
        self.embed_dims = embed_dims
        self.input_dims = num_sample * coords_dim
        self.return_points_embed = return_points_embed
        self.pos_fc = MLPStack(*linear_relu_ln(embed_dims, 1, 2, self.input_dims))

    def forward(self, anchor: torch.Tensor):
        embed = self.pos_fc(anchor)
        return (embed, None) if self.return_points_embed else embed


@POSITIONAL_ENCODING.register_module()
class KeyPoint3DEncoder(BaseModule):
    """Per-instance and per-point embeddings of a poly-line (reference map/blocks.py:45-77)."""

    def __init__(self, embed_dims: int = 256, num_sample: int = 6, coords_dim: int = 2):
        super().__init__()
        self.embed_dims, self.coords_dim, self.num_sample = embed_dims, coords_dim, num_sample
        self.input_dims = num_sample * coords_dim
        self.embed_points = MLPStack(*linear_relu_ln(embed_dims, 1, 2, coords_dim))
        self.embed_instance = MLPStack(*linear_relu_ln(embed_dims, 1, 2, self.input_dims))

    def forward(self, anchor: torch.Tensor):
        bs, num_anchor, _ = anchor.shape
        pts = anchor.reshape(bs, num_anchor * self.num_sample, self.coords_dim)[..., :2]
        return self.embed_instance(anchor), self.embed_points(pts)


@PLUGIN_LAYERS.register_module()
class SparsePoint3DRefinementModule(BaseModule):
    """Residual update of all poly-line coordinates + class logits (reference map/blocks.py:80-135)."""

    def __init__(self, embed_dims: int = 256, num_sample: int = 20, coords_dim: int = 2, num_cls: int = 3,
                 with_cls_branch: bool = True, with_line_key_points=False):
        super().__init__()
        self.embed_dims, self.num_sample, self.num_cls = embed_dims, num_sample, num_cls
        self.with_line_key_points = with_line_key_points
        self.output_dim = coords_dim if with_line_key_points else num_sample * coords_dim
        self.layers = MLPStack(*linear_relu_ln(embed_dims, 2, 2), Linear(embed_dims, self.output_dim),
                                    Scale([1.0] * self.output_dim))
        self.with_cls_branch = with_cls_branch
        if with_cls_branch:
            self.cls_layers = MLPStack(*linear_relu_ln(embed_dims, 1, 2), Linear(embed_dims, num_cls))

    def init_weight(self):
        if self.with_cls_branch:
            nn.init.constant_(self.cls_layers[-1].bias, bias_init_with_prob(0.01))

    def forward(self, instance_feature, anchor, anchor_embed, time_interval=1.0, return_cls=True):
        from hipad_amd import chain as CH
        if return_cls and not self.with_cls_branch:
            raise AssertionError("Without classification layers !!!")
        if CH.usable(instance_feature):
            # both stacks as ONE chain launch; input sum and the residual anchor inside the kernel
            specs = [CH.spec_of(self.layers)] + ([CH.spec_of(self.cls_layers)] if return_cls else [])
            if all(sp is not None for sp in specs):
                calls = [CH.Call(specs[0], instance_feature, anchor_embed, residual=anchor)]
                if return_cls:
                    calls.append(CH.Call(specs[1], instance_feature))
                outs = CH.run(calls)
                return outs[0], (outs[1] if return_cls else None), None
        output = self.layers(instance_feature + anchor_embed) + anchor
        cls = self.cls_layers(instance_feature) if return_cls else None
        return output, cls, None


@PLUGIN_LAYERS.register_module()
class SparsePoint3DKeyPointsGenerator(BaseModule):
    """Key points of a poly-line query: every (x, y) sample is copied to ``len(fix_height)`` heights
    above the ground plane, each with ``num_learnable_pts`` learned planar offsets
    (reference map/blocks.py:137-225)."""

    def __init__(self, embed_dims: int = 256, num_sample: int = 20, num_learnable_pts: int = 0,
                 fix_height: Tuple = (0,), ground_height: int = 0, with_points_embed: bool = False,
                 with_anchor_embed: bool = False):
        super().__init__()
        self.embed_dims, self.num_sample, self.num_learnable_pts = embed_dims, num_sample, num_learnable_pts
        self.with_points_embed, self.with_anchor_embed = with_points_embed, with_anchor_embed
        per_sample = len(fix_height) * num_learnable_pts
        self.num_pts = per_sample if with_points_embed else num_sample * per_sample
        if num_learnable_pts > 0:
            self.learnable_fc = Linear(embed_dims, self.num_pts * 2)
        self.fix_height = np.array(fix_height)
        self.ground_height = ground_height
        # z of every height copy, float32(ground) + float32(offset) like the reference's tensor sum;
        # a non-persistent buffer (not in state_dict) so the forward does no host->device copy
        z = torch.full((len(fix_height),), float(ground_height)) + torch.tensor(self.fix_height, dtype=torch.float32)
        self.register_buffer("_height_levels", z, persistent=False)

    def init_weight(self):
        if self.num_learnable_pts > 0:
            xavier_init(self.learnable_fc, distribution="uniform", bias=0.0)

    def forward(self, anchor, anchor_embed=None, instance_feature=None, T_cur2temp_list=None, cur_timestamp=None,
                temp_timestamps=None):
        if self.num_learnable_pts <= 0:
            raise AssertionError("No learnable pts")
        bs, num_anchor, _ = anchor.shape
        S, Hn, K = self.num_sample, len(self.fix_height), self.num_learnable_pts
        if self.with_anchor_embed:
            if self.with_points_embed:
                src = instance_feature.repeat(1, S, 1) + anchor_embed
            else:
                src = instance_feature + anchor_embed
        else:
            src = instance_feature
        offset = self.learnable_fc(src).reshape(bs, num_anchor, S, Hn, K, 2)
        xy = anchor.reshape(bs, num_anchor, S, 1, 1, -1) + offset
        heights = self._height_levels.to(xy.dtype).reshape(1, 1, 1, Hn, 1, 1)
        z = heights.expand(bs, num_anchor, S, Hn, K, 1)
        key_points = torch.cat([xy, z], dim=-1).flatten(2, 4)
        if cur_timestamp is None or temp_timestamps is None or T_cur2temp_list is None or len(temp_timestamps) == 0:
            return key_points
        warped = []
        for T_cur2temp in T_cur2temp_list[: len(temp_timestamps)]:
            T = T_cur2temp.to(key_points.dtype)[:, None, None]
            warped.append((T[..., :3, :3] @ key_points[..., None]).squeeze(-1) + T[..., :3, 3])
        return key_points, warped

    def project(self, anchor, anchor_embed, instance_feature, projection_mat, image_wh=None):
        """Key points of ``forward`` projected into every camera in the aggregation op's layout
        (bs, A, num_pts, cams, 2): the Linear, then ONE kernel (hipad_line_points_project_*)."""
        from hipad_amd import functional as HF
        if self.num_learnable_pts <= 0 or self.with_points_embed:
            raise NotImplementedError("fused projection covers the configuration the HiP-AD configs use")
        src = instance_feature + anchor_embed if self.with_anchor_embed else instance_feature
        offset = self.learnable_fc(src)
        return HF.line_points_project(anchor, offset, self._height_levels, projection_mat, image_wh, self.num_sample,
                                      len(self.fix_height), self.num_learnable_pts)

    def anchor_projection(self, anchor, T_src2dst_list, src_timestamp=None, dst_timestamps=None, time_intervals=None):
        """Move the (x, y) samples of each poly-line into other ego frames (reference map/blocks.py:227-265)."""
        moved = []
        for T in T_src2dst_list:
            bs, num_anchor, _ = anchor.shape
            T = T.to(anchor.dtype)[:, None]
            pts = anchor.reshape(bs, num_anchor * self.num_sample, -1)
            pts = (T[..., :2, :2] @ pts[..., None]).squeeze(-1) + T[..., :2, 3]
            moved.append(pts.reshape(bs, num_anchor, -1))
        return moved
