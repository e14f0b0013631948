"""3D-box query blocks: anchor encoder, refinement head, key-point generator.

Registered names / constructor keywords / parameter names follow the reference's
``projects/mmdet3d_plugin/models/det/blocks.py`` (encoder :22-74, refinement :77-156, key points
:159-300) so configs and checkpoints carry over; the code is written for this repo.
"""
import torch
import torch.nn as nn

from hipad_amd.compat import (MLPStack, PLUGIN_LAYERS, POSITIONAL_ENCODING, BaseModule, Linear, Scale, bias_init_with_prob,
                              xavier_init)
from projects.mmdet3d_plugin.core.box3d import COS_YAW, H, L, SIN_YAW, VX, W, X, Y, Z

from ..blocks import linear_relu_ln

__all__ = ["SparseBox3DRefinementModule", "SparseBox3DKeyPointsGenerator", "SparseBox3DEncoder"]


@POSITIONAL_ENCODING.register_module()
class SparseBox3DEncoder(BaseModule):
    """Embeds (xyz | log-size | sin,cos yaw | velocity) with one small MLP each and adds or
    concatenates the parts (reference det/blocks.py:22-74)."""

    def __init__(self, embed_dims, vel_dims=3, mode="add", output_fc=True, in_loops=1, out_loops=2):
        super().__init__()
        if mode not in ("add", "cat"):
            raise ValueError(mode)
        self.embed_dims, self.vel_dims, self.mode = embed_dims, vel_dims, mode
        widths = list(embed_dims) if isinstance(embed_dims, (list, tuple)) else [embed_dims] * 5

        def mlp(n_in, n_out):
            return MLPStack(*linear_relu_ln(n_out, in_loops, out_loops, n_in))

        self.pos_fc = mlp(3, widths[0])
        self.size_fc = mlp(3, widths[1])
        self.yaw_fc = mlp(2, widths[2])
        if vel_dims > 0:
            self.vel_fc = mlp(vel_dims, widths[3])
        self.output_fc = mlp(widths[-1], widths[-1]) if output_fc else None

    def forward(self, box_3d: torch.Tensor):
        from hipad_amd import chain as CH
        if self.mode == "cat" and CH.usable(box_3d):
            out = self._forward_chains(box_3d)
            if out is not None:
                return out
        parts = [self.pos_fc(box_3d[..., X:Z + 1]), self.size_fc(box_3d[..., W:H + 1]),
                 self.yaw_fc(box_3d[..., SIN_YAW:COS_YAW + 1])]
        if self.vel_dims > 0:
            parts.append(self.vel_fc(box_3d[..., VX:VX + self.vel_dims]))
        out = torch.cat(parts, dim=-1) if self.mode == "cat" else sum(parts[1:], parts[0])
        return out if self.output_fc is None else self.output_fc(out)

    def _forward_chains(self, box_3d):
        """GPU path: the four part encoders as ONE chain launch writing the column ranges of one tensor ("cat" mode)."""
        from hipad_amd import chain as CH
        fcs = [(self.pos_fc, X, Z + 1), (self.size_fc, W, H + 1), (self.yaw_fc, SIN_YAW, COS_YAW + 1)]
        if self.vel_dims > 0:
            fcs.append((self.vel_fc, VX, VX + self.vel_dims))
        specs = [CH.spec_of(m) for m, _, _ in fcs]
        if any(sp is None for sp in specs):
            return None
        total = sum(sp.N_out for sp in specs)
        calls, col = [], 0
        shape = tuple(box_3d.shape[:-1]) + (total,)
        for sp, (_, lo, hi) in zip(specs, fcs):
            # columns lo..hi of the anchors in, columns col.. of the embedding out: no slice / cat kernels either way
            calls.append(CH.Call(sp, box_3d, x0_cols=(lo, hi), out_slot=CH.OutSlot("embed", shape, col0=col)))
            col += sp.N_out
        (out,) = CH.run(calls)
        return out if self.output_fc is None else self.output_fc(out)


@PLUGIN_LAYERS.register_module()
class SparseBox3DRefinementModule(BaseModule):
    """Residual box update + class logits (+ centerness/yawness quality) from a query
    (reference det/blocks.py:77-156)."""

    def __init__(self, embed_dims=256, output_dim=11, num_cls=10, normalize_yaw=False, refine_yaw=False,
                 with_cls_branch=True, with_quality_estimation=False):
        super().__init__()
        self.embed_dims, self.output_dim, self.num_cls = embed_dims, output_dim, num_cls
        self.normalize_yaw, self.refine_yaw = normalize_yaw, refine_yaw
        self.refine_state = [X, Y, Z, W, L, H] + ([SIN_YAW, COS_YAW] if refine_yaw else [])
        self.layers = MLPStack(*linear_relu_ln(embed_dims, 2, 2), Linear(embed_dims, output_dim),
                                    Scale([1.0] * output_dim))
        self.with_cls_branch = with_cls_branch
        if with_cls_branch:
            self.cls_layers = MLPStack(*linear_relu_ln(embed_dims, 1, 2), Linear(embed_dims, num_cls))
        self.with_quality_estimation = with_quality_estimation
        if with_quality_estimation:
            self.quality_layers = MLPStack(*linear_relu_ln(embed_dims, 1, 2), Linear(embed_dims, 2))

    def init_weight(self):
        if self.with_cls_branch:
            nn.init.constant_(self.cls_layers[-1].bias, bias_init_with_prob(0.01))

    def _fused_update(self, bs, time_interval, like):
        """Per-column factor c with output = delta * c + anchor when the refinement is a pure column-wise affine update
        of the anchor: all state columns refined residually (refine_yaw), no yaw normalisation, one sample (one frame
        interval for every row).  None otherwise."""
        n_state = len(self.refine_state)
        if self.normalize_yaw or n_state != VX or bs != 1:
            return None
        if self.output_dim <= 8:
            return like.new_ones(self.output_dim)
        if not isinstance(time_interval, torch.Tensor):
            time_interval = like.new_tensor(time_interval)
        inv_dt = (1.0 / time_interval.reshape(-1)[:1].to(like.dtype)).expand(self.output_dim - VX)
        return torch.cat([like.new_ones(VX), inv_dt])

    def forward(self, instance_feature, anchor, anchor_embed, time_interval=1.0, return_cls=True):
        from hipad_amd import chain as CH
        cls = quality = grouped = None
        fused = False
        if return_cls and not self.with_cls_branch:
            raise AssertionError("Without classification layers !!!")
        if CH.usable(instance_feature):
            # the regression, class and quality stacks as ONE chain launch (the input sum inside the kernel)
            specs = [CH.spec_of(self.layers)]
            if return_cls:
                specs.append(CH.spec_of(self.cls_layers))
                if self.with_quality_estimation:
                    specs.append(CH.spec_of(self.quality_layers))
            if all(sp is not None for sp in specs):
                col = self._fused_update(instance_feature.shape[0], time_interval, instance_feature)
                if col is not None and specs[0].scale is not None:
                    # output = delta * c + anchor with delta = stack(...) * Scale: fold c into the scale, the anchor in as
                    # the residual -- the slice / add / divide / cat kernels of the update (and their backward) vanish
                    fused = True
                    calls = [CH.Call(specs[0], instance_feature, anchor_embed, residual=anchor, scale=specs[0].scale * col)]
                else:
                    calls = [CH.Call(specs[0], instance_feature, anchor_embed)]
                if len(specs) > 1:
                    calls.append(CH.Call(specs[1], instance_feature))
                if len(specs) > 2:
                    calls.append(CH.Call(specs[2], instance_feature, anchor_embed))
                grouped = CH.run(calls)
        if grouped is not None:
            delta = grouped[0]
            cls = grouped[1] if len(grouped) > 1 else None
            quality = grouped[2] if len(grouped) > 2 else None
            if fused:
                return delta, cls, quality
        else:
            feature = instance_feature + anchor_embed
            delta = self.layers(feature)
        n_state = len(self.refine_state)  # the refined columns are the leading ones: X..H (+ yaw)
        head = delta[..., :n_state] + anchor[..., :n_state]
        pieces = [head]
        if n_state < SIN_YAW + 2:  # yaw columns not refined residually: raw prediction
            pieces.append(delta[..., n_state:COS_YAW + 1])
        if self.normalize_yaw:
            full = torch.cat(pieces, dim=-1)
            yaw = torch.nn.functional.normalize(full[..., SIN_YAW:COS_YAW + 1], dim=-1)
            pieces = [full[..., :SIN_YAW], yaw]
        if self.output_dim > 8:
            if not isinstance(time_interval, torch.Tensor):
                time_interval = instance_feature.new_tensor(time_interval)
            # predicted translation over the frame interval -> velocity increment
            dt = time_interval.reshape(-1, *([1] * (delta.dim() - 1))) if time_interval.dim() else time_interval
            pieces.append(delta[..., VX:] / dt + anchor[..., VX:])
        output = torch.cat(pieces, dim=-1)
        if return_cls and grouped is None:
            cls = self.cls_layers(instance_feature)
            if self.with_quality_estimation:
                quality = self.quality_layers(feature)
        return output, cls, quality


@PLUGIN_LAYERS.register_module()
class SparseBox3DKeyPointsGenerator(BaseModule):
    """Key points of a box query: fixed offsets (fractions of the box size) plus learnable ones,
    rotated by the box yaw and moved to its centre (reference det/blocks.py:159-224)."""

    def __init__(self, embed_dims=256, num_learnable_pts=0, fix_scale=None):
        super().__init__()
        self.embed_dims, self.num_learnable_pts = embed_dims, num_learnable_pts
        if fix_scale is None:
            fix_scale = ((0.0, 0.0, 0.0),)
        self.fix_scale = nn.Parameter(torch.tensor(fix_scale), requires_grad=False)
        self.num_pts = len(self.fix_scale) + num_learnable_pts
        if num_learnable_pts > 0:
            self.learnable_fc = Linear(embed_dims, num_learnable_pts * 3)

    def init_weight(self):
        if self.num_learnable_pts > 0:
            xavier_init(self.learnable_fc, distribution="uniform", bias=0.0)

    def forward(self, anchor, instance_feature=None, T_cur2temp_list=None, cur_timestamp=None,
                temp_timestamps=None):
        bs, num_anchor = anchor.shape[:2]
        size = anchor[..., None, W:H + 1].exp()                      # (bs, A, 1, 3)
        offsets = self.fix_scale * size                                # (bs, A, n_fix, 3)
        if self.num_learnable_pts > 0 and instance_feature is not None:
            learn = self.learnable_fc(instance_feature).reshape(bs, num_anchor, self.num_learnable_pts, 3)
            offsets = torch.cat([offsets, (learn.sigmoid() - 0.5) * size], dim=-2)
        sin, cos = anchor[..., None, SIN_YAW], anchor[..., None, COS_YAW]
        # yaw rotation about z written out (the reference builds a 3x3 matrix and matmuls)
        ox, oy, oz = offsets.unbind(-1)
        key_points = torch.stack([cos * ox - sin * oy, sin * ox + cos * oy, oz], dim=-1) + anchor[..., None, X:Z + 1]
        if cur_timestamp is None or temp_timestamps is None or T_cur2temp_list is None or len(temp_timestamps) == 0:
            return key_points
        # key points warped into earlier frames (ego motion + constant-velocity object motion)
        velocity = anchor[..., VX:]
        warped = []
        for T_cur2temp, t_time in zip(T_cur2temp_list, temp_timestamps):
            dt = (cur_timestamp - t_time).to(velocity.dtype)
            pts = key_points - (velocity * dt[:, None, None])[:, :, None]
            T = T_cur2temp.to(key_points.dtype)[:, None, None]
            warped.append((T[..., :3, :3] @ pts[..., None]).squeeze(-1) + T[..., :3, 3])
        return key_points, warped

    def project(self, anchor, instance_feature, projection_mat, image_wh=None):
        """Key points of ``forward`` projected into every camera, in the aggregation op's location layout
        (bs, A, num_pts, cams, 2) -- ONE kernel (hipad_box_points_project_*) instead of this module's ~15 elementwise
        kernels plus the projection; used by DeformableFeatureAggregation when the inputs are on the GPU."""
        from hipad_amd import functional as HF
        learn = None
        if self.num_learnable_pts > 0 and instance_feature is not None:
            learn = self.learnable_fc(instance_feature)
        elif self.num_learnable_pts > 0:
            raise ValueError("learnable key points need the instance feature")
        return HF.box_points_project(anchor, self.fix_scale, learn, projection_mat, image_wh)

    @staticmethod
    def anchor_projection(anchor, T_src2dst_list, src_timestamp=None, dst_timestamps=None, time_intervals=None):
        """Move box anchors from one ego frame to others (reference det/blocks.py:250-296)."""
        moved = []
        for i, T in enumerate(T_src2dst_list):
            T = T.to(anchor.dtype)[:, None]                            # (bs, 1, 4, 4)
            vel = anchor[..., VX:]
            vdim = vel.shape[-1]
            centre = anchor[..., X:Z + 1]
            if time_intervals is not None:
                dt = time_intervals[i]
            elif src_timestamp is not None and dst_timestamps is not None:
                dt = (src_timestamp - dst_timestamps[i]).to(vel.dtype)
            else:
                dt = None
            if dt is not None:
                centre = centre - vel * dt.reshape(-1, 1, 1)
            centre = (T[..., :3, :3] @ centre[..., None]).squeeze(-1) + T[..., :3, 3]
            cos_sin = (T[..., :2, :2] @ torch.stack([anchor[..., COS_YAW], anchor[..., SIN_YAW]], dim=-1)[..., None]).squeeze(-1)
            vel = (T[..., :vdim, :vdim] @ vel[..., None]).squeeze(-1)
            moved.append(torch.cat([centre, anchor[..., W:H + 1], cos_sin.flip(-1), vel], dim=-1))
        return moved

    @staticmethod
    def distance(anchor):
        return torch.norm(anchor[..., :2], p=2, dim=-1)
