"""Ego-status heads (registered names / keywords / parameter names of the reference's models/ego/blocks.py)."""
import torch.nn as nn

from hipad_amd.compat import PLUGIN_LAYERS, BaseModule, Linear, bias_init_with_prob, linear_relu

from ..blocks import linear_relu_ln

__all__ = ["SparseEgoRefinementModule", "EgoStatusRefinementModule"]


def _mlp3(embed_dims, out_dim):
    return nn.Sequential(*linear_relu(embed_dims, embed_dims), *linear_relu(embed_dims, embed_dims),
                         Linear(embed_dims, out_dim))


@PLUGIN_LAYERS.register_module()
class SparseEgoRefinementModule(BaseModule):
    def __init__(self, embed_dims=256, ego_fut_ts=6, ego_fut_cmd=3, ego_fut_mode=3):
        super().__init__()
        self.embed_dims, self.ego_fut_ts, self.ego_fut_cmd, self.ego_fut_mode = embed_dims, ego_fut_ts, ego_fut_cmd, ego_fut_mode
        self.plan_cls_branch = nn.Sequential(*linear_relu_ln(embed_dims, 1, 2), Linear(embed_dims, 1))
        self.plan_reg_branch = _mlp3(embed_dims, ego_fut_ts * 2)
        self.plan_status_branch = _mlp3(embed_dims, 10)

    def init_weight(self):
        nn.init.constant_(self.plan_cls_branch[-1].bias, bias_init_with_prob(0.01))

    def forward(self, ego_query, ego_feature, ego_anchor_embed):
        bs = ego_query.shape[0]
        cls = self.plan_cls_branch(ego_query).squeeze(-1)
        reg = self.plan_reg_branch(ego_query).reshape(bs, 1, self.ego_fut_cmd * self.ego_fut_mode, self.ego_fut_ts, 2)
        return cls, reg, self.plan_status_branch(ego_feature + ego_anchor_embed)


@PLUGIN_LAYERS.register_module()
class EgoStatusRefinementModule(BaseModule):
    def __init__(self, embed_dims=256, status_dims=6):
        super().__init__()
        self.embed_dims = embed_dims
        self.plan_status_branch = _mlp3(embed_dims, status_dims)

    def forward(self, ego_feature, ego_anchor_embed):
        return self.plan_status_branch(ego_feature + ego_anchor_embed)
