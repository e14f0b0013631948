"""``get_augmentation``: one draw of the per-sequence image augmentation (reference
datasets/bench2drive_dataset.py:709-751, Bench2DriveDataset.get_augmentation).  Host logic; the numpy global RNG is
consumed in the reference's order (resize, bottom crop, horizontal crop, flip, rotate, rotate_3d), so a seeded run
reproduces the reference's aug_config stream."""
import numpy as np

__all__ = ["get_augmentation"]


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
