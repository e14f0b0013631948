"""Ego query store: one fixed box anchor for the ego vehicle, a query feature pooled from the front
camera's coarsest pyramid level, and a one-instance temporal cache.

Registered name, constructor keywords, parameter names (``anchor``, ``ego_feature_encoder.*``) and the
get / cache protocol follow the reference's ``EgoInstanceBank`` (models/ego/instance_bank.py:24-190).
"""
import math

import numpy as np
import torch
from torch import nn

from hipad_amd.compat import MLPStack, PLUGIN_LAYERS, Linear, build_from_cfg
from projects.mmdet3d_plugin.ops import feature_maps_format

from ..blocks import linear_relu_ln
from ..instance_bank import PersistentState, ego_motion_between
from ..plan.instance_bank import front_view_encoder

__all__ = ["EgoInstanceBank"]

# ego box in the lidar frame: centre 0.5 m ahead, resting on the ground plane (-1.84 m), log-size (l, w, h)
_EGO_SIZE = {"nus": (4.08, 1.73, 1.56), "b2d": (4.89, 1.84, 1.49)}


def _ego_anchor(kind):
    length, width, height = _EGO_SIZE[kind]
    return [[0, 0.5, -1.84 + height / 2, math.log(length), math.log(width), math.log(height), 1, 0, 0, 0, 0]]


@PLUGIN_LAYERS.register_module()
class EgoInstanceBank(PersistentState, nn.Module):
    def __init__(self, embed_dims, anchor_type="nus", anchor_handler=None, feature_map_scale=None,
                 num_temp_instances=0, anchor_grad=True, max_time_interval=2, num_anchor=None,
                 with_instance_feat=False, plan_anchor=None, feat_grad=True):
        super().__init__()
        self.embed_dims, self.max_time_interval = embed_dims, max_time_interval
        self.num_temp_instances, self.with_instance_feat = num_temp_instances, with_instance_feat
        if anchor_handler is not None:
            anchor_handler = build_from_cfg(anchor_handler, PLUGIN_LAYERS)
            if not hasattr(anchor_handler, "anchor_projection"):
                raise AssertionError("anchor_handler needs anchor_projection()")
        self.anchor_handler = anchor_handler
        self.anchor = nn.Parameter(torch.tensor(_ego_anchor(anchor_type), dtype=torch.float32), requires_grad=False)
        self.num_anchor = len(self.anchor)
        if with_instance_feat:
            self.instance_feature = nn.Parameter(torch.zeros(self.num_anchor, embed_dims), requires_grad=feat_grad)
        else:
            self.ego_feature_encoder = front_view_encoder(embed_dims, feature_map_scale)
        if plan_anchor is not None:
            self.plan_anchor = nn.Parameter(torch.tensor(np.load(plan_anchor), dtype=torch.float32), requires_grad=False)
            self.plan_anchor_encoder = MLPStack(*linear_relu_ln(embed_dims, 1, 1), Linear(embed_dims, embed_dims))
        self.reset()

    def reset(self):
        self.cached_feature = self.cached_anchor = None
        self.metas = None
        self._drop_state("feature", "anchor", "timestamp")

    def prepare_ego(self, batch_size, feature_maps):
        if self.with_instance_feat:
            feature = self.instance_feature[None].expand(batch_size, -1, -1).contiguous()
        else:
            front = feature_maps_format(feature_maps, inverse=True)[0][-1][:, 0].float()  # coarsest level, front camera
            feature = self.ego_feature_encoder(front).flatten(1)[:, None]
        return feature, self.anchor[None].expand(batch_size, -1, -1).contiguous()

    def get(self, batch_size, metas, feature_maps, dn_metas=None):
        feature, anchor = self.prepare_ego(batch_size, feature_maps)
        if self._kept("anchor") is None or batch_size != self._kept("anchor").shape[0]:
            return feature, anchor, None, None
        # clones: the persistent buffers are overwritten in cache() before backward runs
        self.cached_feature, self.cached_anchor = self._kept("feature").clone(), self._kept("anchor").clone()
        dt = (metas["timestamp"] - self._kept("timestamp")).to(anchor.dtype)
        self.mask = dt.abs() <= self.max_time_interval
        if self.anchor_handler is not None:
            T = ego_motion_between(self.metas, metas, self.cached_anchor)
            self.cached_anchor = self.anchor_handler.anchor_projection(self.cached_anchor, [T], time_intervals=[-dt])[0]
        return feature, anchor, self.cached_feature, self.cached_anchor

    def cache(self, instance_feature, anchor, metas=None, feature_maps=None):
        if self.num_temp_instances <= 0:
            return
        self.metas = metas
        self.cached_feature = self._keep("feature", instance_feature)
        self.cached_anchor = self._keep("anchor", anchor)
        self._keep("timestamp", metas["timestamp"])
