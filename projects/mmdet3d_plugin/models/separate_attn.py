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


def _pieces(tensor, cumsum):
    """Per-modality views of a (bs, tokens, C) tensor.  ONE split instead of a slice per use: its backward
    is a single concatenation of the piece gradients (a slice's backward is a zero-filled full-size tensor
    plus a copy, per slice, plus the adds that merge them)."""
    if tensor is None:
        return None
    sizes = [int(cumsum[i + 1]) - int(cumsum[i]) for i in range(len(cumsum) - 1)]
    return list(torch.split(tensor, sizes, dim=1))


def _gather(pieces, idx):
    if pieces is None:
        return None
    return pieces[idx[0]] if len(idx) == 1 else torch.cat([pieces[i] for i in idx], dim=1)


def _run(idx):
    """(first, last + 1) when the modality indices of a group are consecutive, else None."""
    idx = list(idx)
    return (idx[0], idx[-1] + 1) if idx == list(range(idx[0], idx[-1] + 1)) else None


def _partition(runs, n):
    """The runs, sorted, plus the gaps between them as runs of their own -- or None when two runs overlap."""
    out, at = [], 0
    for i0, i1 in sorted(runs):
        if i0 < at:
            return None
        if i0 > at:
            out.append((at, i0))
        out.append((i0, i1))
        at = i1
    if at < n:
        out.append((at, n))
    return out


def _split_runs(tensor, cumsum, part):
    """{run: view} of a (bs, tokens, C) tensor split ONCE at the run boundaries."""
    if tensor is None:
        return None
    sizes = [int(cumsum[i1]) - int(cumsum[i0]) for i0, i1 in part]
    return dict(zip(part, torch.split(tensor, sizes, dim=1)))


def _views(tensor, cumsum, runs, n):
    """{run: view} for the (possibly overlapping) runs the key groups name: one split when they tile, else one split
    at modality granularity plus a plain slice for every run that spans several modalities."""
    if tensor is None:
        return None
    part = _partition(set(runs), n)
    if part is not None:
        return _split_runs(tensor, cumsum, part)
    out = _split_runs(tensor, cumsum, [(i, i + 1) for i in range(n)])
    for i0, i1 in set(runs):
        if (i0, i1) not in out:
            out[(i0, i1)] = tensor[:, int(cumsum[i0]):int(cumsum[i1])]
    return out


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

    def _indices(self, names):
        return [self.query_select.index(n) for n in names]

    def _route(self, query, key, value, query_pos, key_pos, q_cumsum, k_cumsum, fc_before, fc_after,
               read_updated=False, attn_mask=None):
        """Groups whose modalities are neighbours in the token order (all groups of the HiP-AD configs: [plan, ego],
        [det, map]) read ONE piece of a split made at group granularity -- no gather copies of queries, positional
        embeddings, keys and values, and a backward of one concatenation per split tensor."""
        if attn_mask is not None:
            raise NotImplementedError("attention masks are not used by the HiP-AD configs")
        n = len(self.query_select)
        self_attend = key is None
        q_runs = [_run(self._indices(g)) for g in self._qgroups]
        k_runs = None if self_attend else [_run(self._indices(g)) for g in self._kgroups]
        same_parent = (not self_attend) and key is query
        wanted = set(q_runs) | (set(k_runs) if same_parent else set())
        part_q = None if None in wanted else _partition(wanted, n)
        if part_q is None or (k_runs is not None and None in k_runs):
            return self._route_pieces(query, key, value, query_pos, key_pos, q_cumsum, k_cumsum, fc_before, fc_after,
                                      read_updated)
        qd, qposd = _split_runs(query, q_cumsum, part_q), _split_runs(query_pos, q_cumsum, part_q)
        same_kv = (not self_attend) and value is key
        if self_attend:
            kd = kposd = None
            vd = None if value is None else (qd if value is query else _split_runs(value, q_cumsum, part_q))
        else:
            if same_parent:
                kd, kposd = qd, (qposd if key_pos is query_pos else _views(key_pos, k_cumsum, k_runs, n))
            else:
                kd, kposd = _views(key, k_cumsum, k_runs, n), _views(key_pos, k_cumsum, k_runs, n)
            vd = None if value is None else (kd if same_kv else _views(value, k_cumsum, k_runs, n))
        out_d = dict(qd)        # runs that no group writes keep the input tokens
        for g, attn in enumerate(self.attns):
            run = q_runs[g]
            q, qpos = (out_d if read_updated else qd)[run], (None if qposd is None else qposd[run])
            if self_attend:
                k = kpos = None
                v_in = None if vd is None else vd[run]
            else:
                kr = k_runs[g]
                k, kpos = kd[kr], (None if kposd is None else kposd[kr])
                v_in = None if vd is None else vd[kr]
                if k.shape[1] == 0:  # nothing cached for these modalities: attend within the queries
                    k = kpos = None
                if v_in is not None and v_in.shape[1] == 0:
                    v_in = None
            if self.decouple_list[g]:
                q = torch.cat([q, qpos], dim=-1)
                if k is not None:
                    k = torch.cat([k, kpos], dim=-1)
                if v_in is not None:
                    v_in = fc_before(v_in)
                out_d[run] = fc_after(attn(query=q, key=k, value=v_in, query_pos=None, key_pos=None))
            else:
                out_d[run] = attn(query=q, key=k, value=v_in, query_pos=qpos, key_pos=kpos)
        return torch.cat([out_d[r] for r in part_q], dim=1)

    def _route_pieces(self, query, key, value, query_pos, key_pos, q_cumsum, k_cumsum, fc_before, fc_after,
                      read_updated=False):
        """General form (a group may name any subset of modalities): per-modality pieces, gathered per group."""
        self_attend = key is None
        q_parts, qpos_parts = _pieces(query, q_cumsum), _pieces(query_pos, q_cumsum)
        same_kv = (not self_attend) and value is key
        k_parts = None if self_attend else (q_parts if key is query else _pieces(key, k_cumsum))
        kpos_parts = None if self_attend else (qpos_parts if key_pos is query_pos else _pieces(key_pos, k_cumsum))
        if value is None:
            v_parts = None
        elif self_attend:
            v_parts = q_parts if value is query else _pieces(value, q_cumsum)
        else:
            v_parts = k_parts if same_kv else _pieces(value, k_cumsum)
        out_parts = list(q_parts)  # modality slots that no group writes keep the input tokens
        for g, attn in enumerate(self.attns):
            qi = self._indices(self._qgroups[g])
            q = _gather(out_parts if read_updated else q_parts, qi)
            qpos = _gather(qpos_parts, qi)
            if self_attend:
                k = kpos = None
                v_in = _gather(v_parts, qi)
            else:
                ki = self._indices(self._kgroups[g])
                k, kpos, v_in = _gather(k_parts, ki), _gather(kpos_parts, ki), _gather(v_parts, ki)
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
            if len(qi) == 1:
                out_parts[qi[0]] = out
            else:
                for i, piece in zip(qi, torch.split(out, [q_parts[i].shape[1] for i in qi], dim=1)):
                    out_parts[i] = piece
        return torch.cat(out_parts, dim=1)


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
