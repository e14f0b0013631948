"""Agent-motion head of the unified decoder.

Keeps the registered name, constructor keywords and the two child-module names of the reference head
(models/motion/blocks.py:16-50) so configs build it and checkpoints load; the layers themselves come from the
shared builders in ``..blocks`` / ``hipad_amd.compat`` (Linear + ReLU fused in the MFMA kernel's epilogue).
"""
import torch.nn as nn

from hipad_amd.compat import PLUGIN_LAYERS, BaseModule, bias_init_with_prob

from ..blocks import mlp_head, score_head

__all__ = ["SparseMotionRefinementModule"]


@PLUGIN_LAYERS.register_module()
class SparseMotionRefinementModule(BaseModule):
    """(bs, anchors, modes, C) mode queries -> per-mode score and ``fut_ts`` way-point offsets."""

    def __init__(self, embed_dims=256, fut_ts=12, fut_mode=6):
        super().__init__()
        self.embed_dims = embed_dims
        self.fut_ts = fut_ts
        self.fut_mode = fut_mode
        self.motion_cls_branch = score_head(embed_dims)
        self.motion_reg_branch = mlp_head(embed_dims, 2 * fut_ts)

    def init_weight(self):
        last = self.motion_cls_branch[-1]
        nn.init.constant_(last.bias, bias_init_with_prob(0.01))

    def forward(self, motion_query):
        from hipad_amd import chain as CH
        lead = motion_query.shape[:2]
        specs = (CH.spec_of(self.motion_cls_branch), CH.spec_of(self.motion_reg_branch)) if CH.usable(motion_query) else (None,)
        if all(sp is not None for sp in specs):  # both heads as ONE chain launch
            scores, steps = CH.run([CH.Call(specs[0], motion_query), CH.Call(specs[1], motion_query)])
            return scores[..., 0], steps.reshape(*lead, self.fut_mode, self.fut_ts, 2)
        scores = self.motion_cls_branch(motion_query)[..., 0]
        steps = self.motion_reg_branch(motion_query)
        return scores, steps.reshape(*lead, self.fut_mode, self.fut_ts, 2)
