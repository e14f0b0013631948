"""``get_augmentation``: one draw of the per-sequence image augmentation (reference
datasets/bench2drive_dataset.py:709-751, Bench2DriveDataset.get_augmentation).  Host logic; the numpy global RNG is
consumed in the reference's order (resize, bottom crop, horizontal crop, flip, rotate, rotate_3d), so a seeded run
reproduces the reference's aug_config stream."""
import numpy as np

__all__ = ["get_augmentation", "invert_pose", "camera_matrices"]


def get_augmentation(data_aug_conf, test_mode=False, rng=None):
    if data_aug_conf is None:
        return None
    rng = np.random if rng is None else rng
    H, W = data_aug_conf["H"], data_aug_conf["W"]
    fH, fW = data_aug_conf["final_dim"]
    if not test_mode:
        resize = rng.uniform(*data_aug_conf["resize_lim"])
        new_w, new_h = int(W * resize), int(H * resize)
        crop_h = int((1 - rng.uniform(*data_aug_conf["bot_pct_lim"])) * new_h) - fH
        crop_w = int(rng.uniform(0, max(0, new_w - fW)))
        flip = bool(data_aug_conf["rand_flip"] and rng.choice([0, 1]))
        rotate = rng.uniform(*data_aug_conf["rot_lim"])
        rotate_3d = rng.uniform(*data_aug_conf["rot3d_range"])
    else:
        resize = max(fH / H, fW / W)
        new_w, new_h = int(W * resize), int(H * resize)
        crop_h = int((1 - np.mean(data_aug_conf["bot_pct_lim"])) * new_h) - fH
        crop_w = int(max(0, new_w - fW) / 2)
        flip, rotate, rotate_3d = False, 0, 0
    return {
        "resize": resize,
        "resize_dims": (new_w, new_h),
        "crop": (crop_w, crop_h, crop_w + fW, crop_h + fH),
        "flip": flip,
        "rotate": rotate,
        "rotate_3d": rotate_3d,
    }


def invert_pose(pose):
    """Inverse of a rigid 4x4 pose by transposing the rotation (reference bench2drive_dataset.py:298-302)."""
    inv = np.eye(4)
    inv[:3, :3] = pose[:3, :3].T
    inv[:3, -1] = -inv[:3, :3] @ pose[:3, -1]
    return inv


def camera_matrices(info, data_root=""):
    """The per-camera matrices of one annotation record (reference bench2drive_dataset.py:763-805,
    Bench2DriveDataset.get_data_info): for every ``CAM*`` sensor, in record order,
    ``lidar2img = K_pad @ inv(cam2ego) @ lidar2ego`` etc.  Returns the same keys the reference puts into the sample."""
    import os.path as osp
    sensors = info["sensors"]
    lidar2ego = sensors["LIDAR_TOP"]["lidar2ego"]
    out = dict(img_filename=[], ego2img=[], lidar2img=[], lidar2cam=[], cam_intrinsic=[],
               lidar2global=invert_pose(sensors["LIDAR_TOP"]["world2lidar"]))
    for name, cam in sensors.items():
        if "CAM" not in name:
            continue
        k = cam["intrinsic"]
        k_pad = np.eye(4)
        k_pad[:k.shape[0], :k.shape[1]] = k
        ego2cam = invert_pose(cam["cam2ego"])
        lidar2cam = ego2cam @ lidar2ego
        out["img_filename"].append(osp.join(data_root, cam["data_path"]))
        out["ego2img"].append(k_pad @ ego2cam)
        out["lidar2img"].append(k_pad @ lidar2cam)
        out["lidar2cam"].append(lidar2cam.T)
        out["cam_intrinsic"].append(k_pad)
    return out
