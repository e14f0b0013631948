"""Multi-head attention module of the decoder on the HIP kernel.

Registered name, constructor keywords, forward signature and parameter names
(``attn.in_proj_weight``, ``attn.in_proj_bias``, ``attn.out_proj.{weight,bias}``) follow the
reference's ``MultiheadFlashAttention`` / ``FlashMHA`` (models/attention.py:146-289, 101-143) so its
configs and checkpoints carry over.  The attention core is ``hipad_attention_forward/backward``
(hip-ad_amd/csrc/attn.hip) instead of the flash-attn CUDA package; the packed in-projection is one
GEMM when query, key and value share their input and three otherwise.
"""
import math
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from hipad_amd import chain as CH
from hipad_amd import functional as HF
from hipad_amd.compat import ATTENTION, BaseModule, Linear, build_dropout

__all__ = ["MultiheadFlashAttention", "FlashMHA", "gen_sineembed_for_position"]


class FlashMHA(nn.Module):
    """Packed-projection multi-head attention (parameter layout of reference attention.py:101-143)."""

    def __init__(self, embed_dim, num_heads, bias=True, batch_first=True, attention_dropout=0.0, causal=False,
                 device=None, dtype=None, **kwargs):
        super().__init__()
        if not batch_first:
            raise ValueError("FlashMHA is batch-first")
        if causal:
            raise NotImplementedError("causal attention is not used by the decoder")
        if embed_dim % num_heads:
            raise ValueError("embed_dim must be divisible by num_heads")
        self.embed_dim, self.num_heads, self.causal, self.bias = embed_dim, num_heads, causal, bias
        self.head_dim = embed_dim // num_heads
        if self.head_dim not in (32, 64, 128):
            raise ValueError(f"head_dim {self.head_dim} unsupported by the HIP attention kernel (32/64/128)")
        self.attention_dropout = attention_dropout
        self._seed = HF.new_call_site_seed()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim)) if bias else None
        if not bias:
            self.register_parameter("in_proj_bias", None)
        self.out_proj = Linear(embed_dim, embed_dim, bias=bias)
        self._reset_parameters()

    def _reset_parameters(self):
        nn.init.xavier_uniform_(self.in_proj_weight)
        if self.in_proj_bias is not None:
            nn.init.constant_(self.in_proj_bias, 0.0)
            nn.init.constant_(self.out_proj.bias, 0.0)

    def _project(self, q, k, v):
        E = self.embed_dim
        W, b = self.in_proj_weight, self.in_proj_bias
        if q is k and k is v:
            return HF.linear(q, W, b).split(E, dim=-1)
        if q is k:
            qp, kp = HF.linear(q, W, b, rows=(0, 2 * E)).split(E, dim=-1)   # one split: its backward is one cat
            return qp, kp, HF.linear(v, W, b, rows=(2 * E, 3 * E))
        return (HF.linear(q, W, b, rows=(0, E)), HF.linear(k, W, b, rows=(E, 2 * E)),
                HF.linear(v, W, b, rows=(2 * E, 3 * E)))

    def _project_chains(self, q, k, v, q_pos, k_pos):
        """The three projections (positional inputs added on the fly) as ONE grouped launch of single-layer chains on
        the 256-row blocks of the packed weight: contiguous q / k / v outputs, two backward launches."""
        E = self.embed_dim
        specs = getattr(self, "_hipad_proj_specs", None)
        if specs is None:
            specs = self._hipad_proj_specs = [
                CH.ChainSpec([CH._L(weight=self.in_proj_weight, bias=self.in_proj_bias, rows=(i * E, (i + 1) * E))]) for i in range(3)]
        if k is q and k_pos is q_pos and q_pos is not None:
            q = k = q + q_pos          # shared input: one add feeds both projections (its backward is free)
            q_pos = k_pos = None
        return CH.run([CH.Call(specs[0], q, q_pos), CH.Call(specs[1], k, k_pos), CH.Call(specs[2], v)])

    def forward(self, q, k, v, key_padding_mask=None, q_pos=None, k_pos=None):
        """``q_pos`` / ``k_pos`` (ours, optional): positional tensors to be added to q / k before their projections."""
        if key_padding_mask is not None:
            raise NotImplementedError("key_padding_mask is not used by the decoder")
        if (self.embed_dim <= CH.MAX_WIDTH and self.embed_dim % 16 == 0 and CH.usable(q) and q.dtype == torch.float32
                and self.in_proj_weight.dtype == torch.float32):
            qp, kp, vp = self._project_chains(q, k, v, q_pos, k_pos)
        else:
            if q_pos is not None:
                shared = k is q and k_pos is q_pos
                q = q + q_pos
                k = q if shared else k
                k_pos = None if shared else k_pos
            if k_pos is not None:
                k = k + k_pos
            qp, kp, vp = self._project(q, k, v)
        p_drop = self.attention_dropout if self.training else 0.0
        ctx = HF.attention(qp, kp, vp, self.num_heads, scale=1.0 / math.sqrt(self.head_dim), p_drop=p_drop,
                           seed=self._seed)
        return self.out_proj(ctx), None


@ATTENTION.register_module()
class MultiheadFlashAttention(BaseModule):
    """Attention with positional inputs and an identity connection (reference attention.py:146-289)."""

    def __init__(self, embed_dims, num_heads, attn_drop=0.0, proj_drop=0.0,
                 dropout_layer=dict(type="Dropout", drop_prob=0.0), init_cfg=None, batch_first=True,
                 residual_mode=None, **kwargs):
        super().__init__(init_cfg)
        dropout_layer = dict(dropout_layer) if dropout_layer else dropout_layer
        if "dropout" in kwargs:  # legacy spelling used by the HiP-AD configs
            warnings.warn("`dropout` is deprecated: use attn_drop / proj_drop / dropout_layer", DeprecationWarning)
            attn_drop = kwargs.pop("dropout")
            dropout_layer["drop_prob"] = attn_drop
        self.embed_dims, self.num_heads, self.batch_first = embed_dims, num_heads, True
        self.attn = FlashMHA(embed_dim=embed_dims, num_heads=num_heads, attention_dropout=attn_drop, **kwargs)
        self.residual_mode = residual_mode
        self.proj_drop = nn.Dropout(proj_drop)
        self.dropout_layer = build_dropout(dropout_layer) if dropout_layer else nn.Identity()

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_pos=None, attn_mask=None,
                key_padding_mask=None, **kwargs):
        if attn_mask is not None:
            raise AssertionError("attn mask not supported now.")
        self_keys = key is None
        if self_keys:
            key = query
        if value is None:
            value = key
        if identity is None:
            identity = query
        if key_pos is None and query_pos is not None:
            if query_pos.shape == key.shape:
                key_pos = query_pos
            else:
                warnings.warn(f"position encoding of key is missing in {self.__class__.__name__}.")
        # the positional adds are left to the projection (fused into its kernel where the chain kernels apply)
        out = self.attn(q=query, k=key, v=value, key_padding_mask=key_padding_mask, q_pos=query_pos, k_pos=key_pos)[0]
        if (self.residual_mode != "concat" and isinstance(self.dropout_layer, nn.Dropout) and self.proj_drop.p == 0.0
                and out.is_cuda):
            if not hasattr(self, "_drop_seed"):
                self._drop_seed = HF.new_call_site_seed()
            return HF.dropout_add(out, identity, self.dropout_layer.p, self._drop_seed, self.training)       # one launch
        out = self.dropout_layer(self.proj_drop(out))
        if self.residual_mode == "concat":
            return torch.cat([identity, out], dim=2)
        return identity + out


def gen_sineembed_for_position(pos_tensor, hidden_dim=256):
    """2-D sine/cosine embedding of (x, y) positions, y block first (reference attention.py:292-306)."""
    half = hidden_dim // 2
    idx = torch.arange(half, dtype=torch.float32, device=pos_tensor.device)
    freq = 10000 ** (2 * torch.div(idx, 2, rounding_mode="floor") / half)

    def embed(coord):
        ang = (coord * (2 * math.pi))[..., None] / freq
        return torch.stack((ang[..., 0::2].sin(), ang[..., 1::2].cos()), dim=-1).flatten(-2)

    return torch.cat((embed(pos_tensor[..., 1]), embed(pos_tensor[..., 0])), dim=-1)
