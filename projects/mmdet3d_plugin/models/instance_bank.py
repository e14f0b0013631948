"""Query store of the det / map branches: learnable anchors + features, and the temporal cache
(top-k by decayed confidence, ego-motion warp into the current frame).

Registered name, constructor keywords, parameter names (``anchor``, ``instance_feature``) and the
get / update / cache protocol follow the reference's ``InstanceBank`` (models/instance_bank.py:26-229).
The cache is plain Python state (not in ``state_dict``), per rank, as there.
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from hipad_amd.compat import PLUGIN_LAYERS, build_from_cfg, discrete
from projects.mmdet3d_plugin.core.box3d import VX

__all__ = ["InstanceBank", "select_topk", "ego_motion_between"]


def select_topk(confidence, k, *tensors):
    """Top-k along dim 1 of ``confidence`` (bs, N); gathers the same rows from each (bs, N, ...) tensor."""
    conf, idx = torch.topk(confidence, k, dim=1)
    forced = discrete("topk", idx)
    if forced is not idx:  # a parity test replays another run's selection
        idx, conf = forced, torch.gather(confidence, 1, forced)
    picked = []
    for t in tensors:
        flat = t.reshape(t.shape[0], t.shape[1], -1)
        picked.append(torch.gather(flat, 1, idx[..., None].expand(-1, -1, flat.shape[-1])))
    return conf, picked


def ego_motion_between(prev_metas, metas, like):
    """(bs, 4, 4) transform taking coordinates of the cached frame into the current frame.

    Taken from ``metas["T_temp2cur"]`` when the data side provides it as a device tensor (no host work
    in the step: needed for hipGraph replay); otherwise built from the per-sample ``T_global`` /
    ``T_global_inv`` numpy matrices like the reference does (instance_bank.py:99-104)."""
    if "T_temp2cur" in metas:
        return metas["T_temp2cur"].to(like.dtype)
    mats = [cur["T_global_inv"] @ prev["T_global"] for prev, cur in zip(prev_metas["img_metas"], metas["img_metas"])]
    return like.new_tensor(np.stack(mats))


class PersistentState:
    """Temporal state kept in buffers that are allocated once and then updated IN PLACE, so a captured
    hipGraph of the step keeps reading and writing the same addresses on every replay."""

    def _keep(self, name, value):
        buf = getattr(self, "_p_" + name, None)
        value = value.detach()
        if buf is None or buf.shape != value.shape or buf.dtype != value.dtype or buf.device != value.device:
            buf = value.clone()
            setattr(self, "_p_" + name, buf)
        else:
            buf.copy_(value)
        return buf

    def _kept(self, name):
        return getattr(self, "_p_" + name, None)

    def _drop_state(self, *names):
        for n in names:
            setattr(self, "_p_" + n, None)


@PLUGIN_LAYERS.register_module()
class InstanceBank(PersistentState, nn.Module):
    def __init__(self, num_anchor, embed_dims, anchor, anchor_handler=None, num_temp_instances=0,
                 default_time_interval=0.5, confidence_decay=0.6, anchor_grad=True, feat_grad=True,
                 max_time_interval=2, class_names=None, zero_velocity_classes=None):
        super().__init__()
        self.embed_dims = embed_dims
        self.num_temp_instances = num_temp_instances
        self.default_time_interval = default_time_interval
        self.confidence_decay = confidence_decay
        self.max_time_interval = max_time_interval
        self.class_names, self.zero_velocity_classes = class_names, zero_velocity_classes
        if anchor_handler is not None:
            anchor_handler = build_from_cfg(anchor_handler, PLUGIN_LAYERS)
            if not hasattr(anchor_handler, "anchor_projection"):
                raise AssertionError("anchor_handler needs anchor_projection()")
        self.anchor_handler = anchor_handler
        table = np.load(anchor) if isinstance(anchor, str) else np.asarray(anchor)
        if table.ndim == 3:  # poly-lines: (N, samples, 2) -> (N, samples*2)
            table = table.reshape(table.shape[0], -1)
        self.num_anchor = min(len(table), num_anchor)
        table = table[:num_anchor]
        self.anchor = nn.Parameter(torch.tensor(table, dtype=torch.float32), requires_grad=anchor_grad)
        self.anchor_init = table
        self.instance_feature = nn.Parameter(torch.zeros(self.anchor.shape[0], embed_dims), requires_grad=feat_grad)
        self.reset()

    def init_weight(self):
        self.anchor.data = self.anchor.data.new_tensor(self.anchor_init)
        if self.instance_feature.requires_grad:
            nn.init.xavier_uniform_(self.instance_feature.data, gain=1)

    def reset(self):
        self.cached_feature = self.cached_anchor = None
        self.metas = self.mask = None
        self.confidence = self.temp_confidence = None
        self.instance_id = None
        self.prev_id = 0
        self._drop_state("feature", "anchor", "confidence", "timestamp")

    # ---- per-frame protocol ---------------------------------------------------------------
    def get(self, batch_size, metas=None, dn_metas=None):
        feature = self.instance_feature[None].expand(batch_size, -1, -1).contiguous()
        anchor = self.anchor[None].expand(batch_size, -1, -1).contiguous()
        if self._kept("anchor") is not None and batch_size == self._kept("anchor").shape[0]:
            # frame-local views of the persistent state (the warped anchors are a new tensor)
            # clones: the persistent buffers are overwritten in cache() before backward runs
            self.cached_feature, self.cached_anchor = self._kept("feature").clone(), self._kept("anchor").clone()
            self.confidence = self._kept("confidence").clone()
            dt = (metas["timestamp"] - self._kept("timestamp")).to(feature.dtype)
            self.mask = dt.abs() <= self.max_time_interval
            if self.anchor_handler is not None:
                T = ego_motion_between(self.metas, metas, self.cached_anchor)
                self.cached_anchor = self.anchor_handler.anchor_projection(self.cached_anchor, [T], time_intervals=[-dt])[0]
                if dn_metas is not None and batch_size == dn_metas["dn_anchor"].shape[0]:
                    groups, per = dn_metas["dn_anchor"].shape[1:3]
                    moved = self.anchor_handler.anchor_projection(dn_metas["dn_anchor"].flatten(1, 2), [T],
                                                                  time_intervals=[-dt])[0]
                    dn_metas["dn_anchor"] = moved.reshape(batch_size, groups, per, -1)
            dt = torch.where((dt != 0) & self.mask, dt, torch.full_like(dt, self.default_time_interval))
        else:
            self.reset()
            dt = feature.new_full((batch_size,), self.default_time_interval)
        return feature, anchor, self.cached_feature, self.cached_anchor, dt

    def update(self, instance_feature, anchor, confidence):
        """After the single-frame layer: keep the cached queries and fill up with the best current ones."""
        if self.cached_feature is None:
            return instance_feature, anchor
        extra = instance_feature.shape[1] - self.num_anchor  # denoising queries ride at the end
        tail = None
        if extra > 0:
            tail = (instance_feature[:, -extra:], anchor[:, -extra:])
            instance_feature, anchor, confidence = (t[:, : self.num_anchor] for t in (instance_feature, anchor, confidence))
        fresh = self.num_anchor - self.num_temp_instances
        _, (top_feature, top_anchor) = select_topk(confidence.max(dim=-1).values, fresh, instance_feature, anchor)
        merged_feature = torch.cat([self.cached_feature, top_feature], dim=1)
        merged_anchor = torch.cat([self.cached_anchor, top_anchor], dim=1)
        usable = self.mask[:, None, None]
        instance_feature = torch.where(usable, merged_feature, instance_feature)
        anchor = torch.where(usable, merged_anchor, anchor)
        self.confidence = torch.where(self.mask[:, None], self.confidence, torch.zeros_like(self.confidence))
        if self.instance_id is not None:
            self.instance_id = torch.where(self.mask[:, None], self.instance_id, torch.full_like(self.instance_id, -1))
        if tail is not None:
            instance_feature = torch.cat([instance_feature, tail[0]], dim=1)
            anchor = torch.cat([anchor, tail[1]], dim=1)
        return instance_feature, anchor

    def cache(self, instance_feature, anchor, confidence, metas=None, feature_maps=None):
        if self.num_temp_instances <= 0:
            return
        instance_feature, anchor, confidence = instance_feature.detach(), anchor.detach(), confidence.detach()
        self.metas = metas
        score, label = confidence.max(dim=-1)
        score = score.sigmoid()
        if self.confidence is not None:
            n = self.num_temp_instances
            score = torch.cat([torch.maximum(self.confidence * self.confidence_decay, score[:, :n]), score[:, n:]], dim=1)
        self.temp_confidence = score
        conf, (kept_feature, kept_anchor) = select_topk(score, self.num_temp_instances, instance_feature, anchor)
        if self.class_names and self.zero_velocity_classes is not None:
            _, (kept_label,) = select_topk(score, self.num_temp_instances, label[..., None])
            kept_label = kept_label[..., 0]
            static = torch.zeros_like(kept_label, dtype=torch.bool)
            for name in self.zero_velocity_classes:
                static |= kept_label == self.class_names.index(name)
            kept_anchor = torch.cat([kept_anchor[..., :VX],
                                     torch.where(static[..., None], torch.zeros_like(kept_anchor[..., VX:]),
                                                 kept_anchor[..., VX:])], dim=-1)
        self.confidence = self._keep("confidence", conf)
        self.cached_feature = self._keep("feature", kept_feature)
        self.cached_anchor = self._keep("anchor", kept_anchor)
        self._keep("timestamp", metas["timestamp"])

    # ---- track ids (inference bookkeeping) -------------------------------------------------
    def get_instance_id(self, confidence, anchor=None, threshold=None):
        score = confidence.max(dim=-1).values.sigmoid()
        ids = score.new_full(score.shape, -1).long()
        if self.instance_id is not None and self.instance_id.shape[0] == ids.shape[0]:
            ids[:, : self.instance_id.shape[1]] = self.instance_id
        new = ids < 0
        if threshold is not None:
            new = new & (score >= threshold)
        count = new.sum()
        ids[torch.where(new)] = torch.arange(count).to(ids) + self.prev_id
        self.prev_id += count
        self.update_instance_id(ids, score)
        return ids

    def update_instance_id(self, instance_id=None, confidence=None):
        if self.temp_confidence is None:
            score = confidence.max(dim=-1).values if confidence.dim() == 3 else confidence
        else:
            score = self.temp_confidence
        kept = select_topk(score, self.num_temp_instances, instance_id[..., None])[1][0].squeeze(-1)
        self.instance_id = F.pad(kept, (0, self.num_anchor - self.num_temp_instances), value=-1)
