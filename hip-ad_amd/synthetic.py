"""Synthetic inputs for the hot path: camera rig, projection matrices, pyramid geometry.

Everything here is derived from first principles (a pin-hole rig description), not
copied from the reference; tests/golden/make_golden.py checks that the derived
LIDAR2IMG matrices agree with the constants the reference's closed-loop agent carries
(bench2drive/leaderboard/team_code/hipad_b2d_agent.py:39-67) and that the test-time
resize/crop matrix agrees with datasets/bench2drive_dataset.py:732-741 composed as in
datasets/pipelines/augment.py:27.
"""
import math

import numpy as np

# Bench2Drive sensor rig in the ego frame (x forward, y right, z up), metres / degrees.
# (name, x, y, z, yaw_deg (positive = to the right), horizontal fov_deg)
B2D_RIG = (
    ("CAM_FRONT", 0.80, 0.0, 1.60, 0.0, 70.0),
    ("CAM_FRONT_LEFT", 0.27, -0.55, 1.60, -55.0, 70.0),
    ("CAM_FRONT_RIGHT", 0.27, 0.55, 1.60, 55.0, 70.0),
    ("CAM_BACK", -2.00, 0.0, 1.60, 180.0, 110.0),
    ("CAM_BACK_LEFT", -0.32, -0.55, 1.60, -110.0, 70.0),
    ("CAM_BACK_RIGHT", -0.32, 0.55, 1.60, 110.0, 70.0),
)
B2D_LIDAR_IN_EGO = (-0.39, 0.0, 1.84)
B2D_IMAGE_WH = (1600, 900)


def bench2drive_lidar2img():
    """(6,4,4) float64 lidar->image matrices of the Bench2Drive rig.

    Lidar frame: x right, y forward, z up (origin at the lidar).  Camera frame: x right,
    y down, z along the optical axis.
    """
    W, H = B2D_IMAGE_WH
    mats = []
    for _, ex, ey, ez, yaw, fov in B2D_RIG:
        psi = math.radians(-yaw)  # rotation about z, positive = to the left
        # camera centre in the lidar frame
        cx_l = ey - B2D_LIDAR_IN_EGO[1]
        cy_l = ex - B2D_LIDAR_IN_EGO[0]
        cz_l = ez - B2D_LIDAR_IN_EGO[2]
        right = np.array([math.cos(psi), math.sin(psi), 0.0])
        down = np.array([0.0, 0.0, -1.0])
        fwd = np.array([-math.sin(psi), math.cos(psi), 0.0])
        R = np.stack([right, down, fwd])
        c = np.array([cx_l, cy_l, cz_l])
        l2c = np.eye(4)
        l2c[:3, :3] = R
        l2c[:3, 3] = -R @ c
        f = (W / 2.0) / math.tan(math.radians(fov) / 2.0)
        K = np.eye(4)
        K[0, 0] = K[1, 1] = f
        K[0, 2] = W / 2.0
        K[1, 2] = H / 2.0
        mats.append(K @ l2c)
    return np.stack(mats)


def test_time_aug_matrix(final_hw=(256, 704), src_hw=(900, 1600), bot_pct_lim=(0.0, 0.0)):
    """4x4 image-space matrix of the deterministic test-time resize + crop."""
    fH, fW = final_hw
    H, W = src_hw
    resize = max(fH / H, fW / W)
    newW, newH = int(W * resize), int(H * resize)
    crop_h = int((1 - float(np.mean(bot_pct_lim))) * newH) - fH
    crop_w = int(max(0, newW - fW) / 2)
    m = np.eye(4)
    m[0, 0] = m[1, 1] = resize
    m[0, 3] = -crop_w
    m[1, 3] = -crop_h
    return m


def projection_mats(final_hw=(256, 704), bs=1, dtype=np.float32):
    """(bs,6,4,4) projection_mat and (bs,6,2) image_wh as the data pipeline would emit."""
    m = test_time_aug_matrix(final_hw) @ bench2drive_lidar2img()
    pm = np.broadcast_to(m[None], (bs,) + m.shape).astype(dtype).copy()
    wh = np.broadcast_to(np.array([final_hw[1], final_hw[0]], dtype)[None, None], (bs, 6, 2)).copy()
    return pm, wh


def pyramid_shapes(final_hw=(256, 704), strides=(4, 8, 16, 32)):
    """[(h,w)] per level of the FPN pyramid for an input of final_hw (ceil division like conv stride)."""
    H, W = final_hw
    return [(-(-H // s), -(-W // s)) for s in strides]


def pyramid_tables(final_hw=(256, 704), strides=(4, 8, 16, 32), num_cams=6):
    """spatial_shape (cams,L,2) int32 [h,w] and scale_start_index (cams,L) int32, camera-major."""
    shapes = pyramid_shapes(final_hw, strides)
    ss = np.array([shapes] * num_cams, np.int32)
    sizes = (ss[..., 0] * ss[..., 1]).reshape(-1)
    start = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int32).reshape(num_cams, len(shapes))
    return ss, start, int(sizes.sum())


def ego_motion(step, dx=1.0, dyaw_deg=1.0):
    """T_global (lidar->global) of frame `step`: planar motion, dx metres forward and dyaw per step."""
    yaw = math.radians(dyaw_deg) * step
    T = np.eye(4)
    T[0, 0] = math.cos(yaw)
    T[0, 1] = -math.sin(yaw)
    T[1, 0] = math.sin(yaw)
    T[1, 1] = math.cos(yaw)
    # integrate a gentle arc: position advances along the heading (lidar y is forward)
    pos = np.zeros(2)
    for k in range(step):
        yk = math.radians(dyaw_deg) * k
        pos += dx * np.array([-math.sin(yk), math.cos(yk)])
    T[0, 3], T[1, 3] = pos
    return T


# ------------------------------------------------------------------------------------------
# Query geometry for op-level benchmarks: key points of the stage-2 query sets projected
# through the rig (what the aggregation op sees in the decoder: ~1 of 6 cameras per point).
# ------------------------------------------------------------------------------------------
STAGE2_QUERIES = {  # name: (anchors, points per anchor)  -- SURVEY.md section 8 table
    "det": (900, 13),
    "map": (100, 300),
    "plan": (480, 90),
    "plan48": (48, 90),
    "ego": (1, 13),
}


def _data_dir():
    import os
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "kmeans")


def synthetic_key_points(name, bs=1, seed=0):
    """(bs, A, P, 3) float32 numpy key points in the lidar frame for a stage-2 query set.

    det/ego: box centre + 7 fixed and 6 random offsets scaled by the box size; map: the 20
    poly-line points x 5 heights x 3 jittered copies; plan: 6 way-points x 5 heights x 3 jitter
    for each of the (10 x) 48 trajectory modes.  Geometry only -- the learned offsets of the
    real model are replaced by seeded noise of the same scale.
    """
    import os
    rng = np.random.default_rng(seed)
    A, P = STAGE2_QUERIES[name]
    d = _data_dir()
    if name in ("det", "ego"):
        if name == "det":
            anc = np.load(os.path.join(d, "b2d_det_900.npy")).astype(np.float32)
        else:
            anc = np.array([[0, 0.5, -1.06, np.log(1.9), np.log(4.9), np.log(1.6), 1, 0, 0, 0, 0]], np.float32)
        size = np.exp(anc[:, 3:6])
        fix = np.array([[0, 0, 0], [.45, 0, 0], [-.45, 0, 0], [0, .45, 0], [0, -.45, 0], [0, 0, .45], [0, 0, -.45]], np.float32)
        fix = fix[: min(7, P)]
        learn = rng.uniform(-0.5, 0.5, (A, P - len(fix), 3)).astype(np.float32)
        offs = np.concatenate([np.broadcast_to(fix[None], (A,) + fix.shape), learn], 1) * size[:, None]
        sin, cos = anc[:, 6], anc[:, 7]
        x = cos[:, None] * offs[..., 0] - sin[:, None] * offs[..., 1]
        y = sin[:, None] * offs[..., 0] + cos[:, None] * offs[..., 1]
        kp = np.stack([x, y, offs[..., 2]], -1) + anc[:, None, :3]
    else:
        heights = np.array([0.0, 0.5, 1.0, 1.5, 2.0], np.float32) - 1.84023
        if name == "map":
            pts = np.load(os.path.join(d, "b2d_map_100.npy")).astype(np.float32)  # (100,20,2)
        else:
            base = np.load(os.path.join(d, "b2d_plan_spat_6x8_2m.npy")).astype(np.float32).cumsum(1)  # (48,6,2)
            reps = A // base.shape[0]
            pts = np.concatenate([base * (1.0 + 0.15 * k) for k in range(reps)], 0)
        S = pts.shape[1]
        nl = P // (S * len(heights))
        jit = rng.normal(0, 0.5, (A, S, len(heights), nl, 2)).astype(np.float32)
        xy = pts[:, :, None, None, :] + jit
        z = np.broadcast_to(heights[None, None, :, None, None], xy.shape[:-1] + (1,))
        kp = np.concatenate([xy, z], -1).reshape(A, P, 3)
    return np.broadcast_to(kp[None], (bs,) + kp.shape).astype(np.float32).copy()


def project(key_points, projection_mat, image_wh):
    """numpy float32 restatement of the projection used to build benchmark inputs:
    (bs,A,P,3) -> (bs,A,P,cams,2) normalised image coordinates."""
    kp = np.concatenate([key_points, np.ones_like(key_points[..., :1])], -1)
    p = np.einsum("bcij,bapj->bapci", projection_mat.astype(np.float32), kp.astype(np.float32))
    uv = p[..., :2] / np.maximum(p[..., 2:3], np.float32(1e-5))
    return (uv / image_wh[:, None, None]).astype(np.float32)


# ------------------------------------------------------------------------------------------
# synthetic ground truth for the loss path (20 boxes / 10 poly-lines per sample)
# ------------------------------------------------------------------------------------------
def ground_truth(bs=1, seed=0, n_det=20, n_map=10, num_det_cls=9, num_map_cls=4, ts=6, num_pts=20,
                 input_hw=(256, 704), depth_strides=(4, 8, 16), cams=6, sparse_depth=0.03):
    """Ragged ground truth in the reference's data format (keys of the train pipeline's Collect,
    projects/configs/hipad_b2d_stage2.py:516-523), as CPU torch tensors.  Box counts vary a little per
    sample so the padded path is exercised."""
    import torch
    g = torch.Generator().manual_seed(1000 + seed)
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    u = lambda *s: torch.rand(*s, generator=g)  # noqa: E731
    data = dict(gt_bboxes_3d=[], gt_labels_3d=[], gt_agent_fut_trajs=[], gt_agent_fut_masks=[], gt_map_pts=[],
                gt_map_labels=[])
    for b in range(bs):
        n = max(1, n_det - (b + seed) % 4)
        box = torch.cat([(u(n, 3) * 2 - 1) * torch.tensor([14.0, 28.0, 1.5]), u(n, 3) * 3.0 + 0.6,
                         (u(n, 1) * 2 - 1) * math.pi, r(n, 2) * 3.0], dim=1)
        data["gt_bboxes_3d"].append(box)
        data["gt_labels_3d"].append(torch.randint(0, num_det_cls, (n,), generator=g))
        data["gt_agent_fut_trajs"].append(r(n, ts, 2) * 0.8)
        data["gt_agent_fut_masks"].append((u(n, ts) > 0.2).float())
        m = max(1, n_map - (b + seed) % 3)
        t = torch.linspace(0, 1, num_pts)[None, :, None]
        a, c = (u(m, 1, 2) * 2 - 1) * torch.tensor([13.0, 27.0]), (u(m, 1, 2) * 2 - 1) * torch.tensor([13.0, 27.0])
        line = a + (c - a) * t + r(m, num_pts, 2) * 0.2
        perms = torch.full((m, 2 * (num_pts - 1), num_pts, 2), 1e5)  # VectorizePloyLine(permute=True) padding
        perms[:, 0], perms[:, 1] = line, line.flip(1)
        data["gt_map_pts"].append(perms)
        data["gt_map_labels"].append(torch.randint(0, num_map_cls, (m,), generator=g))
    data["ego_status"] = r(bs, 6)
    data["ego_status_mask"] = torch.ones(bs, 6)
    step = torch.cumsum(u(bs, ts, 1) * torch.tensor([0.1, 1.0]) + torch.tensor([0.0, 0.5]), dim=1)
    for rate, scale in (("2hz", 1.0), ("5hz", 0.4)):
        data[f"gt_ego_fut_trajs_{rate}"] = torch.diff(step * scale, dim=1, prepend=torch.zeros(bs, 1, 2))
        data[f"gt_ego_fut_masks_{rate}"] = torch.ones(bs, ts)
    for rate, scale in (("2m", 0.4), ("5m", 1.0)):
        data[f"gt_ego_spat_trajs_{rate}"] = torch.diff(step * scale, dim=1, prepend=torch.zeros(bs, 1, 2))
        data[f"gt_ego_spat_masks_{rate}"] = torch.ones(bs, ts)
    depth = []
    for s in depth_strides:
        h, w = input_hw[0] // s, input_hw[1] // s
        d = u(bs * cams, h, w) * 50.0 + 1.0
        depth.append(torch.where(u(bs * cams, h, w) < sparse_depth, d, torch.zeros_like(d)))  # LiDAR-sparse
    data["gt_depth"] = depth
    return data
