"""Unified sparse decoder (forward path) on the MI355X kernels.

Registered name ``SparseOneDecoder``, constructor keywords and parameter names follow the reference
(models/sparse_onedecoder.py:35-371) so ``projects/configs/hipad_b2d_stage{1,2}.py`` build it
unchanged and reference checkpoints load.  The forward (reference :472-1092) is reorganised: instead
of one long function with a set of local variables per modality, every query set is a ``_Branch``
record (feature, anchor, embedding, cached counterparts) and the op program
(``concat / temp_gnn / gnn / inter_gnn / norm / split / deformable / ffn / refine``) is a small
dispatcher over those records.  Same arithmetic, same outputs.

Covered: everything the two HiP-AD configs switch on (det / map / plan / ego queries, motion task,
command + target-point embedding, ego status supervision, plan += ego feature, temporal caches,
closed-loop bank rotation).  Not covered (off in those configs, raise at construction): scene
tokens, attention masks, point-level map/plan tokens, top-k mode pruning.
Loss: ``criterion.DecoderLoss`` (static shapes, device-side Hungarian matching).  Post-processing: next row (8f/3).
"""
import copy
from types import SimpleNamespace
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from hipad_amd import functional as HF  # noqa: E402
from hipad_amd.compat import discrete  # noqa: E402
from hipad_amd.compat import (MLPStack, ATTENTION, BBOX_CODERS, BBOX_SAMPLERS, FEEDFORWARD_NETWORK, HEADS, LOSSES, NORM_LAYERS,
                              PLUGIN_LAYERS, POSITIONAL_ENCODING, BaseModule, Linear, build_from_cfg)
from projects.mmdet3d_plugin.core.box3d import COS_YAW, SIN_YAW

from .attention import gen_sineembed_for_position
from .blocks import linear_relu_ln
from .criterion import DecoderLoss

__all__ = ["SparseOneDecoder"]

_UNSUPPORTED_FLAGS = ("with_attn_mask", "with_distance_attn_mask", "with_velocity_attn_mask",
                      "with_plan_group_attn_mask", "with_target_point_next_embed", "with_custom_status_embed",
                      "with_concat_map_points", "with_deform_map_points", "with_concat_plan_points",
                      "with_deform_plan_points", "with_topk_mode")


class _Branch:
    """Per-modality query state while a frame is decoded."""

    __slots__ = ("name", "feature", "anchor", "embed", "points_embed", "temp_feature", "temp_anchor", "temp_embed",
                 "bank", "encoder", "time_interval")

    def __init__(self, name):
        self.name = name
        for k in self.__slots__[1:]:
            setattr(self, k, None)

    @property
    def count(self):
        return self.feature.shape[1]

    @property
    def temp_count(self):
        return 0 if self.temp_feature is None else self.temp_feature.shape[1]


def _rejoin(parts, cut):
    """Concatenate ``parts`` along dim 1 -- or, when they are exactly the views ``cut`` = (whole, views) was split into
    (same objects, same order, nothing missing), return the whole tensor they came from."""
    if cut is not None:
        whole, views = cut
        if (len(parts) == len(views) and all(a is b for a, b in zip(parts, views))
                and sum(v.shape[1] for v in views) == whole.shape[1]):
            return whole
    return torch.cat(parts, dim=1)


_SIDE_STREAMS = {}


def run_concurrently(fns, like):
    """Run independent closures on parallel HIP streams (fork from / join into the current stream) and return
    their results.  The per-modality branches of a decoder layer are chains of tiny kernels (a few workgroups
    each) that do not depend on one another; issued on one stream they execute back to back and the chip idles,
    on separate streams they overlap -- eagerly and, captured, as parallel branches of the hipGraph.  The autograd
    engine runs each backward node on its forward stream, so the backward overlaps the same way.
    ``like``: any tensor of the step (device selection); CPU tensors run the closures in order."""
    if not like.is_cuda or len(fns) <= 1 or not PARALLEL_BRANCHES:
        return [fn() for fn in fns]
    dev = like.device
    pool = _SIDE_STREAMS.setdefault(dev, [])
    while len(pool) < len(fns) - 1:
        pool.append(torch.cuda.Stream(device=dev))
    main = torch.cuda.current_stream(dev)
    results = [None] * len(fns)
    for i, fn in enumerate(fns[1:], start=1):
        side = pool[i - 1]
        side.wait_stream(main)
        with torch.cuda.stream(side):
            results[i] = fn()
    results[0] = fns[0]()
    for i in range(1, len(fns)):
        main.wait_stream(pool[i - 1])
        _record_stream(results[i], main)
    return results


def _record_stream(obj, stream):
    """Tensors made on a side stream are used on ``stream`` from here on: tell the caching allocator."""
    if isinstance(obj, torch.Tensor):
        obj.record_stream(stream)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _record_stream(o, stream)


import os as _os

# Measured on MI355X / ROCm 7.2 (captured step, aggregation op on four streams): 57.2 -> 84.9 ms per frame -- the
# fork/join dependencies between graph branches cost more than the overlap of these short chains returns.  Off.
PARALLEL_BRANCHES = _os.environ.get("HIPAD_PARALLEL_BRANCHES", "0") == "1"


def _optional(cfg, registry, fallback=None):
    """Build ``cfg`` when its type is registered; otherwise ``fallback`` (loss-side components live in
    the "next" rows of SURVEY.md section 8f and are absent from a forward-only install)."""
    if cfg is None:
        return None
    kind = cfg.get("type")
    try:
        return build_from_cfg(cfg, registry)
    except KeyError:
        if fallback is None:
            return None
        return fallback(kind)


@HEADS.register_module()
class SparseOneDecoder(DecoderLoss, BaseModule):
    def __init__(self, ffn: dict = None, init_cfg: dict = None, custom_op: dict = None, embed_dims: int = 256,
                 num_decoder: int = 6, num_single_frame_decoder: int = -1, decouple_attn: bool = True,
                 with_instance_id: bool = True, cls_threshold_to_reg: float = -1, norm_layer: dict = None,
                 graph_model: dict = None, temp_graph_model: dict = None, inter_graph_model: dict = None,
                 operation_order: Optional[List[str]] = None,
                 det_instance_bank: dict = None, map_instance_bank: dict = None, ego_instance_bank: dict = None,
                 plan_instance_bank: dict = None, scenes_instance_bank: dict = None,
                 det_anchor_encoder: dict = None, map_anchor_encoder: dict = None, ego_anchor_encoder: dict = None,
                 plan_anchor_encoder: dict = None,
                 det_deformable: dict = None, map_deformable: dict = None, ego_deformable: dict = None,
                 plan_deformable: dict = None, scenes_attention: dict = None,
                 det_refine_layer: dict = None, map_refine_layer: dict = None, ego_refine_layer: dict = None,
                 plan_refine_layer: dict = None, motion_refine_layer: dict = None, scenes_refine_layer: dict = None,
                 loss_det_cls=None, loss_det_reg=None, loss_map_cls=None, loss_map_reg=None, loss_ego_cls=None,
                 loss_ego_reg=None, loss_ego_status=None, loss_plan_cls=None, loss_plan_reg=None, loss_plan_col=None,
                 loss_plan_dir=None, loss_plan_bound=None, loss_plan_status=None, loss_motion_cls=None,
                 loss_motion_reg=None, loss_scenes_reg=None,
                 det_reg_weights: List = None, map_reg_weights: List = None,
                 det_decoder=None, map_decoder=None, ego_decoder=None, plan_decoder=None, motion_decoder=None,
                 det_sampler=None, map_sampler=None, ego_sampler=None, plan_sampler=None, align_sampler=None,
                 motion_sampler=None,
                 motion_anchor=None, plan_speed_refer=None, plan_anchor_refer=None, combine_layer_loss=True,
                 task_select=("det", "map", "motion", "plan"), query_select=("det", "map", "plan"),
                 num_command=6, independent_gnn=True, independent_temp_gnn=True, independent_inter_gnn=True,
                 with_close_loop=False, open_loop_hz=2, close_loop_hz=20, open_loop_bank_length=None,
                 close_loop_bank_length=None, attn_mask_dict=dict(), with_command_embed=False,
                 with_target_point_embed=False, with_supervise_ego_status=False, with_ego_instance_feature=False,
                 topk_mode_list=None, keep_topk_relative_pos=False, **kwargs):
        super().__init__(init_cfg)
        for flag in _UNSUPPORTED_FLAGS:
            if kwargs.pop(flag, False):
                raise NotImplementedError(f"{flag}=True is not used by the HiP-AD configs and not implemented")
        if scenes_instance_bank is not None or "scenes" in query_select or "scenes" in task_select:
            raise NotImplementedError("scene tokens are not used by the HiP-AD configs")
        if not (independent_gnn and independent_temp_gnn and independent_inter_gnn):
            raise NotImplementedError("only the independent_* = True attention wiring of the configs is implemented")
        self.embed_dims = embed_dims
        self.num_decoder, self.num_single_frame_decoder = num_decoder, num_single_frame_decoder
        self.decouple_attn, self.operation_order = decouple_attn, list(operation_order)
        self.with_instance_id, self.cls_threshold_to_reg = with_instance_id, cls_threshold_to_reg
        self.task_select, self.query_select = list(task_select), list(query_select)
        self.with_command_embed, self.with_target_point_embed = with_command_embed, with_target_point_embed
        self.with_supervise_ego_status, self.with_ego_instance_feature = with_supervise_ego_status, with_ego_instance_feature
        self.with_close_loop, self.open_loop_hz, self.close_loop_hz = with_close_loop, open_loop_hz, close_loop_hz
        self.open_loop_bank_length, self.close_loop_bank_length = open_loop_bank_length, close_loop_bank_length
        self.independent_gnn = self.independent_temp_gnn = self.independent_inter_gnn = True
        self.combine_layer_loss = combine_layer_loss
        self.det_reg_weights, self.map_reg_weights = det_reg_weights, map_reg_weights
        if not self.task_select or not self.query_select:
            raise AssertionError("task_select / query_select must not be empty")

        n_refine = self.operation_order.count("refine")
        n_deform = self.operation_order.count("deformable")

        def stack(cfg, registry, n):
            return nn.ModuleList([build_from_cfg(cfg, registry) for _ in range(n)])

        null_sampler = lambda kind: SimpleNamespace(dn_metas=None, kind=kind)  # noqa: E731
        null_decoder = lambda kind: SimpleNamespace(score_threshold=None, kind=kind)  # noqa: E731
        if "det" in self.query_select:
            self.det_instance_bank = build_from_cfg(det_instance_bank, PLUGIN_LAYERS)
            self.det_anchor_encoder = build_from_cfg(det_anchor_encoder, POSITIONAL_ENCODING)
            self.det_deformable = stack(det_deformable, ATTENTION, n_deform)
            self.det_refine = stack(det_refine_layer, PLUGIN_LAYERS, n_refine)
            self.det_sampler = _optional(det_sampler, BBOX_SAMPLERS, null_sampler) or null_sampler(None)
            self.det_decoder = _optional(det_decoder, BBOX_CODERS, null_decoder) or null_decoder(None)
            self.loss_det_cls, self.loss_det_reg = _optional(loss_det_cls, LOSSES), _optional(loss_det_reg, LOSSES)
        if "map" in self.query_select:
            self.map_instance_bank = build_from_cfg(map_instance_bank, PLUGIN_LAYERS)
            self.map_anchor_encoder = build_from_cfg(map_anchor_encoder, POSITIONAL_ENCODING)
            self.map_deformable = stack(map_deformable, ATTENTION, n_deform)
            self.map_refine = stack(map_refine_layer, PLUGIN_LAYERS, n_refine)
            self.map_sampler = _optional(map_sampler, BBOX_SAMPLERS, null_sampler) or null_sampler(None)
            self.map_decoder = _optional(map_decoder, BBOX_CODERS, null_decoder)
            self.loss_map_cls, self.loss_map_reg = _optional(loss_map_cls, LOSSES), _optional(loss_map_reg, LOSSES)
        if "ego" in self.query_select:
            self.ego_instance_bank = build_from_cfg(ego_instance_bank, PLUGIN_LAYERS)
            if ego_anchor_encoder is not None:
                self.ego_anchor_encoder = build_from_cfg(ego_anchor_encoder, POSITIONAL_ENCODING)
            elif hasattr(self, "det_anchor_encoder"):
                self.ego_anchor_encoder = self.det_anchor_encoder  # shared module, as in the reference
            else:
                self.ego_anchor_encoder = build_from_cfg(det_anchor_encoder, POSITIONAL_ENCODING)
            self.ego_deformable = stack(ego_deformable, ATTENTION, n_deform)
            self.ego_refine = stack(ego_refine_layer, PLUGIN_LAYERS, n_refine)
            self.ego_decoder, self.ego_sampler = _optional(ego_decoder, BBOX_CODERS), _optional(ego_sampler, BBOX_SAMPLERS)
            self.loss_ego_cls, self.loss_ego_reg = _optional(loss_ego_cls, LOSSES), _optional(loss_ego_reg, LOSSES)
            self.loss_ego_status = _optional(loss_ego_status, LOSSES)
        if "plan" in self.query_select:
            self.plan_instance_bank = build_from_cfg(plan_instance_bank, PLUGIN_LAYERS)
            self.plan_anchor_encoder = build_from_cfg(plan_anchor_encoder, POSITIONAL_ENCODING)
            self.plan_deformable = stack(plan_deformable, ATTENTION, n_deform)
            self.plan_refine = stack(plan_refine_layer, PLUGIN_LAYERS, n_refine)
            self.plan_decoder = _optional(plan_decoder, BBOX_CODERS)
            self.plan_sampler, self.align_sampler = _optional(plan_sampler, BBOX_SAMPLERS), _optional(align_sampler, BBOX_SAMPLERS)
            for name, cfg in (("loss_plan_cls", loss_plan_cls), ("loss_plan_reg", loss_plan_reg),
                              ("loss_plan_col", loss_plan_col), ("loss_plan_dir", loss_plan_dir),
                              ("loss_plan_bound", loss_plan_bound), ("loss_plan_status", loss_plan_status)):
                setattr(self, name, _optional(cfg, LOSSES))
            self.ego_fut_ts, self.ego_fut_cmd = self.plan_refine[0].ego_fut_ts, self.plan_refine[0].ego_fut_cmd
            self.ego_fut_mode = self.plan_refine[0].ego_fut_mode
            self.plan_anchor_group = self.plan_instance_bank.anchor_group
            self.plan_anchor_types = self.plan_instance_bank.anchor_types
            self.plan_speed_refer, self.plan_anchor_refer = plan_speed_refer, plan_anchor_refer
            if with_command_embed:
                self.command_embed_encoder = MLPStack(*linear_relu_ln(embed_dims, 2, 1, input_dims=num_command),
                                                           Linear(embed_dims, embed_dims))
            if with_target_point_embed:
                self.target_point_encoder = MLPStack(*linear_relu_ln(embed_dims, 2, 1), Linear(embed_dims, embed_dims))
        if "motion" in self.task_select:
            self.motion_anchor = nn.Parameter(torch.tensor(np.load(motion_anchor), dtype=torch.float32), requires_grad=False)
            self.motion_anchor_encoder = MLPStack(*linear_relu_ln(embed_dims, 1, 1), Linear(embed_dims, embed_dims))
            self.motion_refine = stack(motion_refine_layer, PLUGIN_LAYERS, n_refine)
            self.motion_sampler, self.motion_decoder = _optional(motion_sampler, BBOX_SAMPLERS), _optional(motion_decoder, BBOX_CODERS)
            self.loss_motion_cls, self.loss_motion_reg = _optional(loss_motion_cls, LOSSES), _optional(loss_motion_reg, LOSSES)
            self.fut_ts, self.fut_mode = self.motion_refine[0].fut_ts, self.motion_refine[0].fut_mode

        slot_cfg = {"concat": (custom_op, PLUGIN_LAYERS), "split": (custom_op, PLUGIN_LAYERS),
                    "deformable": (custom_op, PLUGIN_LAYERS), "refine": (custom_op, PLUGIN_LAYERS),
                    "gnn": (graph_model, ATTENTION), "temp_gnn": (temp_graph_model, ATTENTION),
                    "inter_gnn": (inter_graph_model, ATTENTION), "norm": (norm_layer, NORM_LAYERS),
                    "ffn": (ffn, FEEDFORWARD_NETWORK)}
        slots = []
        for op in self.operation_order:
            if op not in slot_cfg:
                raise NotImplementedError(f"{op} is not supported.")
            cfg, registry = slot_cfg[op]
            slots.append(None if cfg is None else build_from_cfg(cfg, registry))
        self.layers = nn.ModuleList(slots)
        if decouple_attn:
            self.fc_before = Linear(embed_dims, embed_dims * 2, bias=False)
            self.fc_after = Linear(embed_dims * 2, embed_dims, bias=False)
        else:
            self.fc_before, self.fc_after = nn.Identity(), nn.Identity()
        self.run_step = 0
        self.is_init_bank_list = False
        # test instrumentation: called as probe(slot, op, {name: live tensor}) after the branches are opened (op "open")
        # and after every op of the program; a probe may overwrite the tensors IN PLACE (teacher forcing under no_grad:
        # tests/test_decoder.py holds every op of the bf16 configuration to the fp32 configuration's inputs)
        self._probe = None

    # ------------------------------------------------------------------------------------
    def init_weights(self):
        for op, layer in zip(self.operation_order, self.layers):
            if layer is None or "refine" in op:
                continue
            for p in layer.parameters():
                if p.dim() > 1:
                    nn.init.xavier_uniform_(p)
        for m in self.modules():
            if hasattr(m, "init_weight"):
                m.init_weight()
        self.init_instance_bank_list()

    def init_instance_bank_list(self):
        """Open loop: one bank per modality.  Closed loop at 20 Hz with a 2 Hz model: a ring of banks so
        each one sees frames 0.5 s apart (reference :396-426)."""
        self.is_init_bank_list = True
        if self.with_close_loop:
            length = self.close_loop_bank_length or self.close_loop_hz // self.open_loop_hz
            rule, clone = self.task_select, copy.deepcopy
        else:
            length = self.open_loop_bank_length or 1
            rule, clone = self.query_select, (lambda bank: bank)
        self.bank_length = length
        for name in ("det", "map", "ego", "plan"):
            if name in rule and hasattr(self, f"{name}_instance_bank"):
                bank = getattr(self, f"{name}_instance_bank")
                setattr(self, f"{name}_instance_bank_list", [clone(bank) for _ in range(length)])

    # ------------------------------------------------------------------------------------
    def get_motion_anchor(self, classification, prediction):
        """Per-box motion-mode anchors of the predicted class, rotated into the lidar frame by the box yaw."""
        modes = self.motion_anchor[discrete("motion_class", classification.argmax(dim=-1))]   # (bs, A, modes, ts, 2)
        box = prediction.detach()
        yaw = torch.atan2(box[..., SIN_YAW], box[..., COS_YAW])
        c, s = yaw.cos()[..., None, None], yaw.sin()[..., None, None]
        x, y = modes[..., 0], modes[..., 1]
        return torch.stack([c * x - s * y, s * x + c * y], dim=-1)

    def _motion_mode_embedding(self, classification, prediction, hidden_dim=256):
        """Sine embedding of the last way-point of every motion-mode anchor (class-conditioned, rotated by the box yaw):
        ``gen_sineembed_for_position(get_motion_anchor(...)[..., -1, :])``, on the GPU as ONE kernel
        (hipad_motion_query_embed) instead of ~25 elementwise launches; no gradient flows through it either way."""
        from hipad_amd import compat as _compat
        if not classification.is_cuda or _compat.discrete_choice[0] is not _compat._identity_choice:
            return gen_sineembed_for_position(self.get_motion_anchor(classification, prediction)[..., -1, :], hidden_dim)
        from hipad_amd import lib as _lib
        half = hidden_dim // 2
        freq = getattr(self, "_sine_freq", None)
        if freq is None or freq.device != classification.device or freq.numel() != half:
            idx = torch.arange(half, dtype=torch.float32, device=classification.device)
            freq = 10000 ** (2 * torch.div(idx, 2, rounding_mode="floor") / half)   # as gen_sineembed_for_position
            self._sine_freq = freq
        with torch.no_grad():
            return _lib.motion_query_embed(classification.detach().float().contiguous(),
                                           prediction.detach().float().contiguous(), self.motion_anchor.detach(), freq,
                                           SIN_YAW, COS_YAW)

    def _frame_constants(self, metas, batch_size):
        """Everything that depends only on the frame's meta data and on parameters, evaluated ONCE per forward instead
        of once per decoder layer (the reference recomputes them in every layer with identical inputs: the values are
        the same, the gradients of the shared encoders arrive as a sum either way):
          * the target-point and command embeddings added to the plan queries' anchor embedding in every refine step
            (reference sparse_onedecoder.py:984-1003);
          * the camera embeddings of all DeformableFeatureAggregation modules (reference blocks.py:187-190) -- on the GPU
            as ONE grouped chain launch for the 24 modules' camera encoders."""
        self._tp_embed = self._cmd_embed = None
        if "plan" in self.task_select:
            if self.with_target_point_embed:
                tp = metas["target_point"].unsqueeze(1).unsqueeze(1)
                self._tp_embed = self.target_point_encoder(gen_sineembed_for_position(tp)).squeeze(1)
            if self.with_command_embed:
                cmd = metas["gt_ego_fut_cmd"].unsqueeze(1).unsqueeze(1)
                self._cmd_embed = self.command_embed_encoder(cmd).squeeze(1)
        mods = [m for n in self.query_select for m in getattr(self, f"{n}_deformable", [])
                if getattr(m, "camera_encoder", None) is not None]
        for m in mods:
            m._cam_embed = None
        pm = metas.get("projection_mat")
        if not mods or pm is None or not pm.is_cuda:
            return
        from hipad_amd import chain as CH
        if not CH.usable(pm):
            return
        specs = [CH.spec_of(m.camera_encoder) for m in mods]
        if any(sp is None for sp in specs):
            return
        cam_in = pm[:, :, :3].reshape(batch_size, pm.shape[1], -1).float()
        outs = CH.run([CH.Call(sp, cam_in) for sp in specs])
        for m, o in zip(mods, outs):
            m._cam_embed = o

    def _open_branches(self, batch_size, metas, feature_maps, bank_idx):
        br = {}
        for name in self.query_select:
            b = br[name] = _Branch(name)
            b.bank = getattr(self, f"{name}_instance_bank_list")[bank_idx]
            b.encoder = getattr(self, f"{name}_anchor_encoder")
            if name in ("ego", "plan"):
                b.feature, b.anchor, b.temp_feature, b.temp_anchor = b.bank.get(batch_size, metas, feature_maps)
            else:
                sampler = getattr(self, f"{name}_sampler")
                b.feature, b.anchor, b.temp_feature, b.temp_anchor, b.time_interval = b.bank.get(
                    batch_size, metas, dn_metas=sampler.dn_metas)
            b.embed = self._encode(b, b.anchor)
            if b.temp_anchor is not None:
                b.temp_embed = self._encode(b, b.temp_anchor, keep_points=False)
        return br

    @staticmethod
    def _encode(b, anchor, keep_points=True):
        out = b.encoder(anchor)
        if isinstance(out, tuple):  # poly-line encoders return (instance embedding, point embedding)
            if keep_points:
                b.points_embed = out[1]
            return out[0]
        return out

    # ------------------------------------------------------------------------------------
    def forward(self, img, feature_maps, metas):
        if isinstance(feature_maps, torch.Tensor):
            feature_maps = [feature_maps]
        if not self.is_init_bank_list:
            self.init_instance_bank_list()
        batch_size = feature_maps[0].shape[0]
        bank_idx = self.run_step % self.bank_length
        br = self._open_branches(batch_size, metas, feature_maps, bank_idx)
        order = self.query_select
        with_temp = any(br[n].temp_anchor is not None for n in order)
        self.num_anchor_list = [br[n].count for n in order]
        self.num_temp_anchor_list = [br[n].temp_count for n in order]
        for n in ("det", "map", "plan", "ego"):
            setattr(self, f"num_{n}_anchor", br[n].count if n in br else 0)
            setattr(self, f"num_temp_{n}_anchor", br[n].temp_count if n in br else 0)
        self.num_anchor_cumsum = np.cumsum([0] + self.num_anchor_list)
        self.num_temp_anchor_cumsum = np.cumsum([0] + self.num_temp_anchor_list)
        self.total_num_anchor, self.total_num_temp_anchor = self.num_anchor_cumsum[-1], self.num_temp_anchor_cumsum[-1]

        outs = {k: dict(classification=[], prediction=[], quality=[], status=[]) for k in ("det", "map", "ego", "plan", "motion")}
        tokens = embeds = temp_tokens = temp_embeds = None
        deform_i = refine_i = 0
        time_interval = next((br[n].time_interval for n in ("map", "det") if n in br), None)
        det_cls = map_cls = plan_cls = None

        def probe(slot, op, extras=None):
            if self._probe is None:
                return
            state = {}
            for k, v in (("tokens", tokens), ("embeds", embeds), ("temp_tokens", temp_tokens), ("temp_embeds", temp_embeds)):
                if v is not None and op not in ("split", "deformable", "refine", "open", "refine.det", "refine.map"):
                    state[k] = v
            if op in ("refine.det", "refine.map"):
                pass
            elif op in ("split", "deformable", "refine", "open"):
                for n in order:
                    for k in ("feature", "anchor", "embed", "temp_feature", "temp_anchor", "temp_embed"):
                        v = getattr(br[n], k)
                        if v is not None:
                            state[f"{n}.{k}"] = v
            for k, v in (extras or {}).items():
                if v is not None:
                    state["out." + k] = v
            self._probe(slot, op, state)

        probe(-1, "open")
        self._frame_constants(metas, batch_size)
        parents = {}     # what the last "split" cut: name -> (whole tensor, the views handed to the branches)
        for slot, (op, layer) in enumerate(zip(self.operation_order, self.layers)):
            if layer is None:
                continue
            extras = None
            if op == "concat":
                # pieces that still ARE the views the last split handed out join back into their parent without a launch
                # (the embeddings and the cached instances do not change between a layer's two concatenations)
                tokens = _rejoin([br[n].feature for n in order], parents.get("feature"))
                embeds = _rejoin([br[n].embed for n in order], parents.get("embed"))
                if with_temp:
                    cached = [br[n] for n in order if br[n].temp_feature is not None]
                    temp_tokens = _rejoin([b.temp_feature for b in cached], parents.get("temp_feature"))
                    temp_embeds = _rejoin([b.temp_embed for b in cached], parents.get("temp_embed"))
            elif op == "split":
                # one split per tensor (backward = one concatenation), not a slice per modality
                sizes = [int(v) for v in self.num_anchor_list]
                fs, es = torch.split(tokens, sizes, dim=1), torch.split(embeds, sizes, dim=1)
                parents["feature"], parents["embed"] = (tokens, fs), (embeds, es)
                for n, f, e in zip(order, fs, es):
                    br[n].feature, br[n].embed = f, e
                if with_temp:
                    tsizes = [int(v) for v in self.num_temp_anchor_list]
                    tfs, tes = [], []
                    for n, c, f, e in zip(order, tsizes, torch.split(temp_tokens, tsizes, dim=1),
                                          torch.split(temp_embeds, tsizes, dim=1)):
                        if c > 0 or br[n].temp_feature is not None:
                            br[n].temp_feature, br[n].temp_embed = f, e
                            tfs.append(f)
                            tes.append(e)
                    parents["temp_feature"], parents["temp_embed"] = (temp_tokens, tfs), (temp_embeds, tes)
            elif op == "temp_gnn":
                tokens = layer(tokens, temp_tokens, temp_tokens, query_pos=embeds, key_pos=temp_embeds,
                               num_anchor_cumsum=self.num_anchor_cumsum,
                               num_temp_anchor_cumsum=self.num_temp_anchor_cumsum if with_temp else None,
                               fc_before=self.fc_before, fc_after=self.fc_after)
            elif op in ("gnn", "inter_gnn"):
                tokens = layer(tokens, None, tokens, query_pos=embeds, num_anchor_cumsum=self.num_anchor_cumsum,
                               fc_before=self.fc_before, fc_after=self.fc_after)
            elif op in ("norm", "ffn"):
                tokens = layer(tokens)
            elif op == "deformable":
                def aggregate(n, i=deform_i):
                    b = br[n]
                    return getattr(self, f"{n}_deformable")[i](b.feature, b.anchor, b.embed, feature_maps, metas)
                # the four modalities' aggregation modules are independent: one stream each
                for n, f in zip(order, run_concurrently([lambda n=n: aggregate(n) for n in order], tokens)):
                    br[n].feature = f
                deform_i += 1
            elif op == "refine":
                layer_no = refine_i + 1
                if "det" in self.task_select:
                    d = br["det"]
                    d.anchor, det_cls, det_qt = self.det_refine[refine_i](d.feature, d.anchor, d.embed,
                                                                          time_interval=time_interval, return_cls=True)
                    # (probe before the class scores drive the temporal top-k merge and the motion-mode choice)
                    probe(slot, "refine.det", {"det.classification": det_cls, "det.prediction": d.anchor, "det.quality": det_qt})
                    outs["det"]["prediction"].append(d.anchor)
                    outs["det"]["classification"].append(det_cls)
                    outs["det"]["quality"].append(det_qt)
                    if layer_no == self.num_single_frame_decoder:
                        d.feature, d.anchor = d.bank.update(d.feature, d.anchor, det_cls)
                    d.embed = self._encode(d, d.anchor)
                    if layer_no > self.num_single_frame_decoder and d.temp_embed is not None:
                        d.temp_embed = d.embed[:, : d.bank.num_temp_instances]
                if "map" in self.task_select:
                    m = br["map"]
                    m.anchor, map_cls, map_qt = self.map_refine[refine_i](m.feature, m.anchor, m.embed,
                                                                          time_interval=time_interval, return_cls=True)
                    probe(slot, "refine.map", {"map.classification": map_cls, "map.prediction": m.anchor})
                    outs["map"]["prediction"].append(m.anchor)
                    outs["map"]["classification"].append(map_cls)
                    outs["map"]["quality"].append(map_qt)
                    if layer_no == self.num_single_frame_decoder:
                        m.feature, m.anchor = m.bank.update(m.feature, m.anchor, map_cls)
                    m.embed = self._encode(m, m.anchor)
                    if layer_no > self.num_single_frame_decoder and m.temp_embed is not None:
                        m.temp_embed = m.embed[:, : m.bank.num_temp_instances]
                if "motion" in self.task_select:
                    d = br["det"]
                    mode_query = self.motion_anchor_encoder(self._motion_mode_embedding(det_cls, d.anchor))
                    motion_cls, motion_reg = self.motion_refine[refine_i](mode_query + (d.feature + d.embed).unsqueeze(2))
                    outs["motion"]["classification"].append(motion_cls)
                    outs["motion"]["prediction"].append(motion_reg)
                if "ego" in self.task_select:
                    g = br["ego"]
                    if not self.with_supervise_ego_status:
                        raise NotImplementedError("ego trajectory head (with_supervise_ego_status=False) is unused")
                    outs["ego"]["classification"].append(None)
                    outs["ego"]["prediction"].append(None)
                    outs["ego"]["status"].append(self.ego_refine[refine_i](g.feature, g.embed))
                if "plan" in self.task_select:
                    p = br["plan"]
                    # the row vectors (bs, 1, C) are broadcast over the plan queries: one launch per sum
                    ego_b = br["ego"] if self.with_ego_instance_feature else None
                    embed = HF.add_rows(p.embed, self._tp_embed if self.with_target_point_embed else None,
                                        self._cmd_embed if self.with_command_embed else None,
                                        ego_b.embed if ego_b is not None else None)
                    if ego_b is not None:
                        p.feature = HF.add_rows(p.feature, ego_b.feature)
                    plan_reg, plan_cls = self.plan_refine[refine_i](p.feature, p.anchor, embed, True)
                    p.anchor = plan_reg
                    bs, nj, _ = plan_reg.shape
                    wp = plan_reg.reshape(bs, 1, nj, self.ego_fut_ts, 2)
                    steps = HF.step_offsets(wp)                                            # way-points -> offsets
                    outs["plan"]["prediction"].append(steps)
                    outs["plan"]["classification"].append(plan_cls.reshape(bs, 1, -1))
                    outs["plan"]["status"].append(None)
                    p.embed = self._encode(p, p.anchor)
                refine_i += 1
                if self._probe is not None:
                    extras = {f"{k}.{f}": (outs[k][f][-1] if outs[k][f] else None)
                              for k in outs for f in ("classification", "prediction", "quality", "status")}
            probe(slot, op, extras)

        det_output = dict(classification=outs["det"]["classification"], prediction=outs["det"]["prediction"],
                          quality=outs["det"]["quality"],
                          instance_feature=br["det"].feature if "det" in br else None,
                          anchor_embed=br["det"].embed if "det" in br else None)
        map_output = dict(classification=outs["map"]["classification"], prediction=outs["map"]["prediction"],
                          quality=outs["map"]["quality"],
                          instance_feature=br["map"].feature if "map" in br else None,
                          anchor_embed=br["map"].embed if "map" in br else None)
        ego_output = dict(classification=outs["ego"]["classification"], prediction=outs["ego"]["prediction"],
                          status=outs["ego"]["status"])
        plan_output = dict(classification=outs["plan"]["classification"], prediction=outs["plan"]["prediction"],
                           status=outs["plan"]["status"])
        motion_output = dict(classification=outs["motion"]["classification"], prediction=outs["motion"]["prediction"])
        scenes_output = dict(scenes_latent_tokens=[], scenes_latent_embeds=[], scenes_future_tokens=[],
                             scenes_future_embeds=[])

        # remember this frame's instances for the next one
        if "ego" in br:
            br["ego"].bank.cache(br["ego"].feature, br["ego"].anchor, metas, feature_maps)
        if "det" in br:
            br["det"].bank.cache(br["det"].feature, br["det"].anchor, det_cls, metas, feature_maps)
        if "map" in br:
            br["map"].bank.cache(br["map"].feature, br["map"].anchor, map_cls, metas, feature_maps)
        if "plan" in br:
            br["plan"].bank.cache(br["plan"].feature, br["plan"].anchor, plan_cls, metas, feature_maps)
        if self.with_instance_id and "det" in self.task_select:
            det_output["instance_id"] = br["det"].bank.get_instance_id(det_cls, br["det"].anchor,
                                                                       self.det_decoder.score_threshold)
        self.run_step += 1
        for n in self.query_select:   # the per-frame camera embeddings must not outlive the forward
            for m in getattr(self, f"{n}_deformable", []):
                m._cam_embed = None
        self._tp_embed = self._cmd_embed = None
        return det_output, map_output, ego_output, plan_output, motion_output, scenes_output

    # ------------------------------------------------------------------------------------
    # loss(): DecoderLoss (criterion.py) -- target assignment and every loss term on the device

    def post_process(self, det_output, map_output, ego_output, plan_output, motion_output, data, output_idx=-1):
        """Per-task result decoding (reference sparse_onedecoder.py:1581-1605)."""
        det = mp = ego = plan = motion = None
        if "det" in self.task_select:
            det = self.det_decoder.decode(det_output["classification"], det_output["prediction"],
                                          det_output.get("instance_id"), det_output.get("quality"), output_idx=output_idx)
        if "map" in self.task_select:
            mp = self.map_decoder.decode(map_output["classification"], map_output["prediction"],
                                         map_output.get("instance_id"), map_output.get("quality"), output_idx=output_idx)
        if "motion" in self.task_select:
            motion = self.motion_decoder.decode(det_output["classification"], det_output["prediction"],
                                                det_output.get("instance_id"), det_output.get("quality"), motion_output)
        if "ego" in self.task_select and not self.with_supervise_ego_status:
            raise NotImplementedError("ego trajectory decoding is unused by the HiP-AD configs")
        if "plan" in self.task_select:
            plan = self.plan_decoder.decode(ego_output, det_output, motion_output, plan_output, data)
        return det, mp, ego, plan, motion
