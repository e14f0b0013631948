"""Routing of the concatenated decoder tokens into per-modality attention groups.

Three registered modules with the reference's names, constructor keywords and ``attns.{i}``
parameter layout (models/separate_attn.py:25-185, 188-331, 334-721):

  SeparateAttention          group g: queries = keys = modalities of ``separate_list[g]``
                             (from the cached tokens when ``key`` is given)
  TemporalSeparateAttention  queries ``query_list[g]`` (current) -> keys ``key_list[g]`` (cached;
                             current tokens when nothing is cached)
  InteractiveAttention       queries ``query_list[g]`` -> keys ``key_list[g]``, both current

All three are one routine here: gather the token ranges of a group, run its attention (with the
decoupled 512-d variant: positional embedding concatenated instead of added, value up-projected by
``fc_before``, result down-projected by ``fc_after``), and write the result back into the query slots.
Additive attention masks (``with_attn_mask`` and the distance / velocity / ban / cancel variants of
InteractiveAttention) are off in the HiP-AD configs and not implemented.
"""
import numpy as np
import torch
import torch.nn as nn

from hipad_amd.compat import ATTENTION, build_from_cfg

__all__ = ["SeparateAttention", "TemporalSeparateAttention", "InteractiveAttention"]


def _build_attns(attn, count):
    if isinstance(attn, dict):
        return nn.Sequential(*[build_from_cfg(attn, ATTENTION) for _ in range(count)])
    if isinstance(attn, (list, tuple)):
        return nn.Sequential(*[build_from_cfg(a, ATTENTION) for a in attn])
    raise NotImplementedError(type(attn))


def _take(tensor, spans):
    """Concatenate token ranges [(start, end), ...] along dim 1 (a view when there is one range)."""
    if tensor is None:
        return None
    parts = [tensor[:, s:e] for s, e in spans]
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)


class _GroupedAttention(nn.Module):
    def _setup(self, attn, query_select, query_groups, key_groups, decouple_list):
        if query_groups is None or decouple_list is None:
            raise AssertionError("group lists are required")
        if not (len(query_groups) == len(key_groups) == len(decouple_list)):
            raise AssertionError("group lists differ in length")
        self.query_select = query_select
        self.decouple_list = decouple_list
        self._qgroups, self._kgroups = query_groups, key_groups
        self.attns = _build_attns(attn, len(query_groups))

    def _spans(self, names, cumsum):
        idx = [self.query_select.index(n) for n in names]
        return [(int(cumsum[i]), int(cumsum[i + 1])) for i in idx]

    def _route(self, query, key, value, query_pos, key_pos, q_cumsum, k_cumsum, fc_before, fc_after,
               read_updated=False, attn_mask=None):
        if attn_mask is not None:
            raise NotImplementedError("attention masks are not used by the HiP-AD configs")
        result = query.clone()
        self_attend = key is None
        for g, attn in enumerate(self.attns):
            qs = self._spans(self._qgroups[g], q_cumsum)
            q = _take(result if read_updated else query, qs)
            qpos = _take(query_pos, qs)
            if self_attend:
                k = v_in = kpos = None
                vs = qs
                v_in = _take(value, vs)
            else:
                ks = self._spans(self._kgroups[g], k_cumsum)
                k, kpos, v_in = _take(key, ks), _take(key_pos, ks), _take(value, ks)
                if k.shape[1] == 0:  # nothing cached for these modalities: attend within the queries
                    k = kpos = None
                if v_in is not None and v_in.shape[1] == 0:
                    v_in = None
            if self.decouple_list[g]:
                q = torch.cat([q, qpos], dim=-1)
                if k is not None:
                    k = torch.cat([k, kpos], dim=-1)
                qpos = kpos = None
                if v_in is not None:
                    v_in = fc_before(v_in)
                out = fc_after(attn(query=q, key=k, value=v_in, query_pos=qpos, key_pos=kpos))
            else:
                out = attn(query=q, key=k, value=v_in, query_pos=qpos, key_pos=kpos)
            off = 0
            for s, e in qs:
                result[:, s:e] = out[:, off:off + (e - s)]
                off += e - s
        return result


@ATTENTION.register_module()
class SeparateAttention(_GroupedAttention):
    def __init__(self, attn=None, embed_dims=256, query_select=None, separate_list=None, decouple_list=None, **kwargs):
        super().__init__()
        self.separate_list = separate_list
        self._setup(attn, query_select, separate_list, separate_list, decouple_list)

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, attn_mask=None,
                num_anchor_cumsum=None, num_temp_anchor_cumsum=None, fc_before=None, fc_after=None, **kwargs):
        return self._route(query, key, value, query_pos, key_pos, num_anchor_cumsum, num_temp_anchor_cumsum,
                           fc_before, fc_after, attn_mask=attn_mask)


@ATTENTION.register_module()
class TemporalSeparateAttention(_GroupedAttention):
    def __init__(self, attn=None, embed_dims=256, query_select=None, query_list=None, key_list=None,
                 decouple_list=None, use_updated_query=False, **kwargs):
        super().__init__()
        self.query_list, self.key_list, self.use_updated_query = query_list, key_list, use_updated_query
        self._setup(attn, query_select, query_list, key_list, decouple_list)

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, attn_mask=None,
                num_anchor_cumsum=None, num_temp_anchor_cumsum=None, fc_before=None, fc_after=None, **kwargs):
        if key is None or num_temp_anchor_cumsum is None:
            # first frame of a sequence: every group looks at the current tokens of its key modalities
            key, key_pos, num_temp_anchor_cumsum = query, query_pos, num_anchor_cumsum
        return self._route(query, key, value, query_pos, key_pos, num_anchor_cumsum, num_temp_anchor_cumsum,
                           fc_before, fc_after, read_updated=self.use_updated_query, attn_mask=attn_mask)


@ATTENTION.register_module()
class InteractiveAttention(_GroupedAttention):
    def __init__(self, attn=None, embed_dims=256, query_select=None, query_list=None, key_list=None,
                 decouple_list=None, with_distance_attn_mask=False, with_velocity_attn_mask=False,
                 attn_mask_ban_list=None, attn_mask_cancel_list=None, **kwargs):
        super().__init__()
        if with_distance_attn_mask or with_velocity_attn_mask or attn_mask_ban_list or attn_mask_cancel_list:
            raise NotImplementedError("distance / velocity / ban / cancel masks are off in the HiP-AD configs")
        self.query_list, self.key_list = query_list, key_list
        self._setup(attn, query_select, query_list, key_list, decouple_list)

    def forward(self, query, key=None, value=None, query_pos=None, key_pos=None, attn_mask=None,
                num_anchor_cumsum=None, num_temp_anchor_cumsum=None, fc_before=None, fc_after=None, **kwargs):
        return self._route(query, query, value, query_pos, query_pos, num_anchor_cumsum, num_anchor_cumsum,
                           fc_before, fc_after, attn_mask=attn_mask)
