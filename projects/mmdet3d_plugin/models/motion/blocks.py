"""Agent-motion head (registered name / keywords / parameter names of the reference's models/motion/blocks.py)."""
import torch.nn as nn

from hipad_amd.compat import PLUGIN_LAYERS, BaseModule, Linear, bias_init_with_prob, linear_relu

from ..blocks import linear_relu_ln

__all__ = ["SparseMotionRefinementModule"]


@PLUGIN_LAYERS.register_module()
class SparseMotionRefinementModule(BaseModule):
    def __init__(self, embed_dims=256, fut_ts=12, fut_mode=6):
        super().__init__()
        self.embed_dims, self.fut_ts, self.fut_mode = embed_dims, fut_ts, fut_mode
        self.motion_cls_branch = nn.Sequential(*linear_relu_ln(embed_dims, 1, 2), Linear(embed_dims, 1))
        self.motion_reg_branch = nn.Sequential(*linear_relu(embed_dims, embed_dims), *linear_relu(embed_dims, embed_dims),
                                               Linear(embed_dims, fut_ts * 2))

    def init_weight(self):
        nn.init.constant_(self.motion_cls_branch[-1].bias, bias_init_with_prob(0.01))

    def forward(self, motion_query):
        bs, num_anchor = motion_query.shape[:2]
        cls = self.motion_cls_branch(motion_query).squeeze(-1)
        reg = self.motion_reg_branch(motion_query).reshape(bs, num_anchor, self.fut_mode, self.fut_ts, 2)
        return cls, reg
