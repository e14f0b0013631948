"""Result decoding of the unified decoder's head outputs (inference / closed loop).

Registered names and constructor keywords of the reference's result decoders -- ``SparseBox3DDecoder``
(models/det/decoder.py:21-107), ``SparsePoint3DDecoder`` (models/map/decoder.py:6-37), ``SparseMotionDecoder``
(models/motion/decoder.py:379-472) and ``SparsePlanDecoder`` (models/plan/decoder.py:62-329, including the
collision-aware rescoring of plan modes against the predicted agent motion) -- and the same result dictionaries
(``boxes_3d / scores_3d / labels_3d / cls_scores / instance_ids``, ``vectors / scores / labels``,
``trajs_3d / trajs_score``, ``plan_{temp,spat}_* / plan_speed_*``).  The closed-loop agent consumes
``plan_speed_5hz`` and ``plan_spat_2m`` (bench2drive/leaderboard/team_code/hipad_b2d_agent.py:564-578).

The ranking / thresholding / rescoring is done with batched tensor ops on the device; only the final per-sample
dictionaries are moved to the host, as the reference does.
"""
import math

import torch

from hipad_amd.compat import BBOX_CODERS
from projects.mmdet3d_plugin.core.box3d import CNS, COS_YAW, SIN_YAW, VX

__all__ = ["decode_box", "SparseBox3DDecoder", "SparsePoint3DDecoder", "SparseMotionDecoder", "SparsePlanDecoder"]


def decode_box(box):
    """(…, [x,y,z, log w,l,h, sin,cos, v…]) -> (…, [x,y,z, w,l,h, yaw, v…])."""
    yaw = torch.atan2(box[..., SIN_YAW], box[..., COS_YAW])
    return torch.cat([box[..., :3], box[..., 3:6].exp(), yaw[..., None], box[..., VX:]], dim=-1)


def _rank_boxes(cls_scores, quality, num_output, score_threshold, sorted_, instance_id):
    """Shared ranking of SparseBox3DDecoder / SparseMotionDecoder: top-``num_output`` (query, class) pairs by class
    score, re-ranked by score * sigmoid(centerness) when the quality head exists.  Returns per-batch tensors
    (scores, scores_before_centerness or None, flat indices, class ids, keep mask or None)."""
    squeeze = instance_id is not None
    scores = cls_scores.sigmoid()
    cls_ids = None
    if squeeze:
        scores, cls_ids = scores.max(dim=-1)
        scores = scores.unsqueeze(-1)
    num_cls = scores.shape[-1]
    scores, indices = scores.flatten(start_dim=1).topk(num_output, dim=1, sorted=sorted_)
    if not squeeze:
        cls_ids = indices % num_cls
    mask = scores >= score_threshold if score_threshold is not None else None
    origin = None
    if quality is not None:
        centerness = torch.gather(quality[..., CNS], 1, indices // num_cls)
        origin = scores.clone()
        scores, order = torch.sort(scores * centerness.sigmoid(), dim=1, descending=True)
        if not squeeze:
            cls_ids = torch.gather(cls_ids, 1, order)
        if mask is not None:
            mask = torch.gather(mask, 1, order)
        indices = torch.gather(indices, 1, order)
    return scores, origin, indices, cls_ids, mask, num_cls, squeeze


@BBOX_CODERS.register_module()
class SparseBox3DDecoder:
    def __init__(self, num_output: int = 300, score_threshold=None, sorted: bool = True):
        self.num_output, self.score_threshold, self.sorted = num_output, score_threshold, sorted

    def decode(self, cls_scores, box_preds, instance_id=None, quality=None, output_idx=-1):
        qt = None if quality is None or quality[output_idx] is None else quality[output_idx]
        scores, origin, indices, cls_ids, mask, num_cls, squeeze = _rank_boxes(
            cls_scores[output_idx], qt, self.num_output, self.score_threshold, self.sorted, instance_id)
        boxes = box_preds[output_idx]
        output = []
        for i in range(scores.shape[0]):
            query = indices[i] // num_cls
            labels = cls_ids[i][indices[i]] if squeeze else cls_ids[i]
            keep = slice(None) if mask is None else mask[i]
            item = dict(boxes_3d=decode_box(boxes[i, query][keep]).cpu(), scores_3d=scores[i][keep].cpu(),
                        labels_3d=labels[keep].cpu())
            if origin is not None:
                # NB the reference gathers `cls_scores_origin` BEFORE the centerness re-ranking and never re-orders
                # it (det/decoder.py:62-66, 86-88): position k holds the k-th best raw class score
                item["cls_scores"] = origin[i][keep].cpu()
            if instance_id is not None:
                item["instance_ids"] = instance_id[i, indices[i]][keep]
            output.append(item)
        return output


@BBOX_CODERS.register_module()
class SparsePoint3DDecoder:
    def __init__(self, coords_dim: int = 2, score_threshold=None):
        self.coords_dim, self.score_threshold = coords_dim, score_threshold

    def decode(self, cls_scores, pts_preds, instance_id=None, quality=None, output_idx=-1):
        bs, num_pred, num_cls = cls_scores[-1].shape
        scores, indices = cls_scores[-1].sigmoid().flatten(start_dim=1).topk(num_pred, dim=1)
        pts = pts_preds[-1].reshape(bs, num_pred, -1, self.coords_dim)
        labels = indices % num_cls
        output = []
        for i in range(bs):
            keep = slice(None) if self.score_threshold is None else scores[i] >= self.score_threshold
            lines = pts[i, indices[i] // num_cls][keep].detach().cpu().numpy()
            output.append(dict(vectors=[v for v in lines], scores=scores[i][keep].detach().cpu().numpy(),
                               labels=labels[i][keep].detach().cpu().numpy()))
        return output


@BBOX_CODERS.register_module()
class SparseMotionDecoder(SparseBox3DDecoder):
    def __init__(self):
        super().__init__()

    def decode(self, cls_scores, box_preds, instance_id=None, quality=None, motion_output=None, output_idx=-1):
        qt = None if quality is None or quality[output_idx] is None else quality[output_idx]
        scores, _, indices, _, mask, num_cls, _ = _rank_boxes(
            cls_scores[output_idx], qt, self.num_output, self.score_threshold, self.sorted, instance_id)
        boxes = box_preds[output_idx]
        trajs, traj_cls = motion_output["prediction"][-1], motion_output["classification"][-1].sigmoid()
        output = []
        for i in range(scores.shape[0]):
            query = indices[i] // num_cls
            keep = slice(None) if mask is None else mask[i]
            centre = decode_box(boxes[i, query][keep])[:, None, None, :2]
            output.append(dict(trajs_3d=(trajs[i, query][keep].cumsum(dim=-2) + centre).cpu(),
                               trajs_score=traj_cls[i, query][keep].cpu()))
        return output


# ------------------------------------------------------------------------------------------------
# planning
# ------------------------------------------------------------------------------------------------
def _bev_corners(box):
    """(N, 7) [x,y,z,w,l,h,yaw] -> (N, 4, 2): the four ground-plane corners the reference takes from
    box3d_to_corners_gpu(...)[:, [0, 3, 7, 4], :2] (datasets/utils.py:31-50)."""
    half = box[:, None, 3:5] * box.new_tensor([[-0.5, -0.5], [-0.5, 0.5], [0.5, 0.5], [0.5, -0.5]])[None]
    c, s = torch.cos(box[:, 6])[:, None], torch.sin(box[:, 6])[:, None]
    x = half[..., 0] * c - half[..., 1] * s
    y = half[..., 0] * s + half[..., 1] * c
    return torch.stack([x, y], dim=-1) + box[:, None, :2]


def _corners_in_box(a, b):
    """Any ground-plane corner of b inside a, in a's frame (plan/decoder.py:25-60; note that the reference compares the
    local x against a's index-3 extent and the local y against its index-4 extent)."""
    yaw, loc = a[:, 6], a[:, :2]
    c, s = torch.cos(-yaw), torch.sin(-yaw)
    rel = b[:, :2] - loc
    local = torch.stack([rel[:, 0] * c - rel[:, 1] * s, rel[:, 0] * s + rel[:, 1] * c], dim=-1)
    b_local = torch.cat([local, b[:, 2:6], (b[:, 6] - yaw)[:, None]], dim=-1)
    corners = _bev_corners(b_local)
    ext_x, ext_y = a[:, 3:4], a[:, 4:5]
    inside = (corners[..., 0] <= ext_x / 2) & (corners[..., 0] >= -ext_x / 2) & \
             (corners[..., 1] <= ext_y / 2) & (corners[..., 1] >= -ext_y / 2)
    return inside.any(dim=-1)


def check_collision(boxes1, boxes2):
    return _corners_in_box(boxes1, boxes2) | _corners_in_box(boxes2, boxes1)


def _with_origin(traj):
    return torch.cat([traj.new_zeros(traj.shape[:-2] + (1, 2)), traj], dim=-2)


def _heading(traj, start_yaw, static_dis_thresh):
    """Heading along a way-point sequence by central differences; sequences shorter than the threshold keep
    the start heading (plan/decoder.py:229-252)."""
    yaw = traj.new_zeros(traj.shape[:-1])
    yaw[..., 1:-1] = torch.atan2(traj[..., 2:, 1] - traj[..., :-2, 1], traj[..., 2:, 0] - traj[..., :-2, 0])
    yaw[..., -1] = torch.atan2(traj[..., -1, 1] - traj[..., -2, 1], traj[..., -1, 0] - traj[..., -2, 0])
    yaw[..., 0] = start_yaw
    static = torch.linalg.norm(traj[..., -1, :] - traj[..., 0, :], dim=-1) < static_dis_thresh
    return torch.where(static.unsqueeze(-1), yaw[..., 0].unsqueeze(-1), yaw)


@BBOX_CODERS.register_module()
class SparsePlanDecoder:
    EGO_SIZE = {"nus": [4.08, 1.73, 1.56], "b2d": [4.89, 1.84, 1.49]}

    def __init__(self, ego_fut_ts=6, ego_fut_cmd=3, ego_fut_mode=3, ego_vehicle="nus", anchor_types=None,
                 anchor_refer=None, speed_refer=None, with_rescore=False, adapt_status=False):
        if anchor_types is None:
            raise AssertionError("anchor_types is required")
        self.ego_fut_ts, self.ego_fut_cmd, self.ego_fut_mode = ego_fut_ts, ego_fut_cmd, ego_fut_mode
        self.ego_size = self.EGO_SIZE[ego_vehicle]
        self.with_rescore, self.adapt_status = with_rescore, adapt_status
        self.anchor_types, self.anchor_refer, self.speed_refer = anchor_types, anchor_refer, speed_refer
        self.num_group = len(anchor_types)

    # ---- agent context shared by the rescoring calls ------------------------------------------------
    @staticmethod
    def _agents(det_output, motion_output):
        det_cls = det_output["classification"][-1].sigmoid()
        return dict(anchors=det_output["prediction"][-1], confidence=det_cls.max(dim=-1).values,
                    motion_cls=motion_output["classification"][-1].sigmoid(),
                    motion_reg=motion_output["prediction"][-1].cumsum(-2))

    def rescore(self, plan_cls, plan_reg, motion_cls, motion_reg, det_anchors, det_confidence, score_thresh=0.15,
                static_dis_thresh=0.5, dim_scale=1.1, num_motion_mode=1, offset=0.5, ego_fut_ts=None, ego_fut_mode=None):
        """plan_cls (bs, M) scores of M ego modes with way-points plan_reg (bs, M, ts, 2): modes whose swept ego box
        meets the most likely predicted trajectory of a confident agent get -999 (unless every mode collides)."""
        ts = self.ego_fut_ts if ego_fut_ts is None else ego_fut_ts
        modes = self.ego_fut_mode if ego_fut_mode is None else ego_fut_mode
        bs = plan_reg.shape[0]
        if bs != det_anchors.shape[0]:
            raise NotImplementedError("plan rescoring runs per sample (batch 1), as in the reference's closed loop")
        ego_xy = _with_origin(plan_reg)
        ego_box = det_anchors.new_zeros(bs, modes, ts + 1, 7)
        ego_box[..., :2] = ego_xy
        ego_box[..., 3:6] = ego_box.new_tensor(self.ego_size) * dim_scale
        ego_box[..., 6] = _heading(ego_xy, math.pi / 2, static_dis_thresh)
        agent_xy = _with_origin(motion_reg[..., :ts, :]) + det_anchors[:, :, None, None, :2]
        best = motion_cls.topk(num_motion_mode, dim=-1).indices[..., None, None].expand(-1, -1, -1, ts + 1, 2)
        agent_xy = torch.gather(agent_xy, 2, best)
        agent_box = agent_xy.new_zeros(agent_xy.shape[:-1] + (7,))
        agent_box[..., :2] = agent_xy
        agent_box[..., 3:6] = det_anchors[..., None, None, 3:6].exp()
        box_yaw = torch.atan2(det_anchors[..., SIN_YAW], det_anchors[..., COS_YAW])
        agent_box[..., 6] = _heading(agent_xy, box_yaw[..., None].expand(agent_xy.shape[:-2]), static_dis_thresh)
        agent_box = torch.where((det_confidence < score_thresh)[:, :, None, None, None], agent_box.new_tensor(1e6), agent_box)
        ego_box, agent_box = ego_box[..., 1:, :], agent_box[..., 1:, :]
        num_anchor, num_motion = agent_box.shape[1:3]
        ego_flat = ego_box[:, None, None].expand(-1, num_anchor, num_motion, -1, -1, -1).reshape(-1, 7).clone()
        agent_flat = agent_box.unsqueeze(3).expand(-1, -1, -1, modes, -1, -1).reshape(-1, 7)
        # the reference nudges the ego boxes forward with `ego_box[0] += offset * cos(ego_box[6])` and
        # `ego_box[1] += offset * sin(ego_box[6])` on the FLATTENED (N, 7) tensor (plan/decoder.py:284-285):
        # rows 0 and 1 receive row 6's cos / sin, nothing else moves.  Kept, for identical rankings.
        if ego_flat.shape[0] > 6:
            row6 = ego_flat[6].clone()
            ego_flat[0] += offset * torch.cos(row6)
            ego_flat[1] += offset * torch.sin(row6)
        col = check_collision(ego_flat, agent_flat).reshape(bs, num_anchor, num_motion, modes, ts)
        col = col.permute(0, 3, 1, 2, 4).flatten(2, -1).any(dim=-1)          # (bs, modes)
        all_col = col.all(dim=-1)
        col = col & ~all_col[:, None]
        return plan_cls + col.float() * -999, all_col

    def select(self, det_output, motion_output, cls_preds, reg_preds, data):
        bs, fut_cmd, _ = cls_preds[0].shape
        rows = torch.arange(bs, device=cls_preds[0].device)
        cmd = data["gt_ego_fut_cmd"].argmax(dim=-1) if fut_cmd > 1 else 0
        cls_preds = [c[rows, cmd] for c in cls_preds]
        reg_preds = [r[rows, cmd] for r in reg_preds]
        can_rescore = self.with_rescore and len(det_output["prediction"]) > 0 and len(motion_output["prediction"]) > 0
        if can_rescore and ("temp", "2hz") in self.anchor_types:
            k = self.anchor_types.index(("temp", "2hz"))
            ag = self._agents(det_output, motion_output)
            cls_preds[k], _ = self.rescore(cls_preds[k], reg_preds[k], ag["motion_cls"], ag["motion_reg"], ag["anchors"],
                                           ag["confidence"], ego_fut_mode=reg_preds[k].shape[1])
        mode = cls_preds[self.anchor_types.index(self.anchor_refer)].argmax(dim=-1)
        return [c[rows, mode] for c in cls_preds], [r[rows, mode] for r in reg_preds]

    def rescore_speed(self, speed, det_output, motion_output):
        ag = self._agents(det_output, motion_output)
        rate = self.speed_refer[1]
        if rate not in speed:
            raise NotImplementedError(rate)
        cls = speed[rate]["cls_preds"].permute(1, 0)
        reg = speed[rate]["reg_preds"].unsqueeze(0)
        motion_reg = ag["motion_reg"]
        if rate == "5hz":  # compare the 5 Hz way-points 3 and 6 with the agents' 2 Hz steps 1 and 2
            reg, motion_reg = reg[:, :, [2, 5]], motion_reg[:, :, :, [0, 1]]
        cls, all_col = self.rescore(cls, reg, ag["motion_cls"], motion_reg, ag["anchors"], ag["confidence"],
                                    ego_fut_ts=reg.shape[2], ego_fut_mode=reg.shape[1])
        for grp in speed.values():
            grp["cls_preds"] = cls.permute(1, 0)
            grp["reg_preds"] = grp["reg_preds"] * (1 - all_col.float())
        return speed

    def decode(self, ego_output, det_output, motion_output, planning_output, data):
        prediction, classification = planning_output["prediction"][-1], planning_output["classification"][-1]
        bs = classification.shape[0]
        cls_preds = [c.reshape(bs, self.ego_fut_cmd, -1) for c in classification.chunk(self.num_group, dim=2)]
        reg_preds = [r.reshape(bs, self.ego_fut_cmd, -1, self.ego_fut_ts, 2).cumsum(dim=-2)
                     for r in prediction.chunk(self.num_group, dim=2)]
        cls_preds, reg_preds = self.select(det_output, motion_output, cls_preds, reg_preds, data)
        can_rescore = self.with_rescore and len(det_output["prediction"]) > 0 and len(motion_output["prediction"]) > 0
        outputs = []
        for i in range(bs):
            out, speed = {}, {}
            for k, kind in enumerate(self.anchor_types):
                if kind[0] in ("temp", "spat"):
                    out[f"plan_{kind[0]}_{kind[1]}"] = reg_preds[k][i].cpu()
                elif kind[0] == "speed":
                    grp = speed.setdefault(kind[1], dict(cls_preds=[], reg_preds=[], speed_areas=[]))
                    grp["cls_preds"].append(cls_preds[k]); grp["reg_preds"].append(reg_preds[k][i]); grp["speed_areas"].append(kind[2])
                else:
                    raise NotImplementedError(kind)
            for grp in speed.values():
                grp["cls_preds"], grp["reg_preds"] = torch.stack(grp["cls_preds"]), torch.stack(grp["reg_preds"])
            if speed and can_rescore:
                speed = self.rescore_speed(speed, det_output, motion_output)
            for rate, grp in speed.items():
                pick = torch.argmax(grp["cls_preds"])
                if self.adapt_status:
                    ego_speed = float(ego_output["status"][-1][i, 0, 0])
                    for k, (lo, hi) in enumerate(grp["speed_areas"]):
                        if lo <= ego_speed < hi:
                            pick = k
                out[f"plan_speed_{rate}"] = grp["reg_preds"][pick].cpu()
            outputs.append(out)
        return outputs
