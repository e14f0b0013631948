"""From stored camera frames to the training step's inputs, in this repo's own terms:

  scene_rotation / rotate_scene      the BEV rotation augmentation as batched 4x4 / 3x3 products on the padded
                                     ground-truth arrays (reference datasets/pipelines/augment.py:95-138 does the same
                                     arithmetic box list by box list inside a pipeline class)
  ADAPT / adapt_sample               the last pipeline step (reference datasets/pipelines/transform.py:107-168) as a TABLE
                                     of derived entries: projection matrices, image sizes, poses, focal lengths, wrapped
                                     yaws, tensors for the ground truth, the channels-first image stack
  SequenceFrames                     a frame source for hipad_amd.frame.GraphedTrainStep / TrainStep that walks driving
                                     sequences with the sequence-grouped sampler (datasets/samplers), draws the image
                                     augmentation per sample (datasets/augmentation.get_augmentation), runs the DEVICE image
                                     pipeline (hipad_amd.imgpipe: Pillow-exact resize / crop / flip / rotate + normalise,
                                     two launches per sample) on uint8 frames resident in HBM and composes the augmented
                                     projection matrices -- what bench.py's `stage2_full_frames` workload times.

Host logic + calls into the device image pipeline; no file IO (decoding JPEGs is out of section 8's scope: the frames are
synthetic uint8 arrays of the camera resolution).
"""
import numpy as np
import torch

from . import synthetic as syn

__all__ = ["scene_rotation", "rotate_scene", "adapt_sample", "SequenceFrames"]


# ---------------------------------------------------------------------------------------------------------------------
# BEV rotation of a scene
# ---------------------------------------------------------------------------------------------------------------------
def scene_rotation(angle):
    """4x4 rotation about the vertical axis by ``angle`` (counter-clockwise seen from above), float64."""
    m = np.eye(4)
    c, s = np.cos(angle), np.sin(angle)
    m[:2, :2] = [[c, -s], [s, c]]
    return m


def rotate_scene(lidar2img, lidar2global, boxes, angle):
    """The scene turned by ``angle`` about the lidar's vertical axis: every sensor matrix absorbs the inverse turn
    (stacked product), box centres and velocities turn with the scene, yaws shift.  lidar2img (n, 4, 4), lidar2global
    (4, 4) or None, boxes (G, >= 7) or None -> the same three, new arrays."""
    undo = scene_rotation(angle).T                      # inverse of a rotation
    turned = np.asarray(lidar2img) @ undo
    pose = None if lidar2global is None else np.asarray(lidar2global) @ undo
    out = None
    if boxes is not None:
        out = np.array(boxes, copy=True)
        spin = scene_rotation(angle)[:3, :3].T          # row vectors: p' = p @ spin
        out[:, :3] = out[:, :3] @ spin
        out[:, 6] = out[:, 6] + angle
        nv = out.shape[1] - 7
        if nv > 0:
            out[:, 7:] = out[:, 7:] @ spin[:nv, :nv]
    return turned, pose, out


# ---------------------------------------------------------------------------------------------------------------------
# the sample -> step-input adaptor as a table
# ---------------------------------------------------------------------------------------------------------------------
def _wrap_yaw(boxes):
    boxes = np.array(boxes, copy=True)
    yaw = boxes[:, 6]
    boxes[:, 6] = yaw - np.floor(yaw / (2 * np.pi) + 0.5) * (2 * np.pi)       # into [-pi, pi)
    return torch.as_tensor(boxes).float()


def _image_stack(img):
    if isinstance(img, torch.Tensor) and img.dim() == 4:
        return img                                                            # already (n, 3, h, w): device pipeline
    if isinstance(img[0], torch.Tensor):
        return torch.stack(list(img), 0).permute(0, 3, 1, 2).contiguous()
    return torch.as_tensor(np.ascontiguousarray(np.stack(img, 0).transpose(0, 3, 1, 2)))


def _as_tensor(x):
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))


# (output key, input keys it needs, function of those inputs); evaluated in order, missing inputs skip the entry
ADAPT = (
    ("projection_mat", ("lidar2img",), lambda m: np.stack(m).astype(np.float32)),
    ("image_wh", ("img_shape",), lambda s: np.ascontiguousarray(np.asarray(s, dtype=np.float32)[:, 1::-1])),
    ("T_global_inv", ("lidar2global",), np.linalg.inv),
    ("T_global", ("lidar2global",), lambda p: p),
    ("cam_intrinsic", ("cam_intrinsic",), lambda k: np.stack(k).astype(np.float32)),
    ("focal", ("cam_intrinsic",), lambda k: k[..., 0, 0]),
    ("instance_id", ("instance_inds",), lambda i: i),
    ("gt_bboxes_3d", ("gt_bboxes_3d",), _wrap_yaw),
    ("gt_labels_3d", ("gt_labels_3d",), lambda v: _as_tensor(v).long()),
    ("img", ("img",), _image_stack),
) + tuple((k, (k,), _as_tensor) for k in ("gt_map_labels", "gt_map_pts", "gt_agent_fut_trajs", "gt_agent_fut_masks",
                                          "gt_ego_fut_trajs", "gt_ego_fut_masks", "gt_ego_fut_cmd", "command_near_xy",
                                          "ego_status"))


def adapt_sample(sample):
    """Apply ADAPT to one sample dict in place (entries whose inputs are absent are skipped) and return it."""
    for key, needs, fn in ADAPT:
        if all(k in sample for k in needs):
            sample[key] = fn(*(sample[k] for k in needs))
    return sample


# ---------------------------------------------------------------------------------------------------------------------
# frame source: sequences -> sampler -> device image pipeline -> step inputs
# ---------------------------------------------------------------------------------------------------------------------
class _ToySequences:
    """``num_seq`` driving sequences of ``seq_len`` frames each (``flag`` = sequence of a frame), the training image
    augmentation of the stage configs, drawn once per sequence (keep_consistent_seq_aug=True,
    projects/configs/hipad_b2d_stage2.py:612)."""
    keep_consistent_seq_aug = True

    def __init__(self, num_seq, seq_len, data_aug_conf, rng):
        self.flag = np.repeat(np.arange(num_seq), seq_len)
        self.seq_len, self.conf, self.rng = seq_len, data_aug_conf, rng

    def __len__(self):
        return len(self.flag)

    def get_augmentation(self):
        from projects.mmdet3d_plugin.datasets.augmentation import get_augmentation
        return get_augmentation(self.conf, test_mode=False, rng=self.rng)


class SequenceFrames:
    """Frame source with the interface of hipad_amd.frame.SyntheticFrames (``bs``, ``device``, ``step``, ``next()``)
    whose images come through the data pipeline (module docstring).  Raw frames: a small pool of uint8 (6, H, W, 3)
    tensors in HBM (synthetic content; one per sequence phase), poses: the planar ego motion of synthetic.ego_motion
    along each sequence, ground truth: SyntheticFrames' padded sets."""

    def __init__(self, bs=1, input_hw=(256, 704), device="cuda", seed=0, num_seq=8, seq_len=40, raw_hw=(900, 1600), pool=4,
                 mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375)):
        from projects.mmdet3d_plugin.datasets.samplers import GroupInBatchSampler
        from .frame import SyntheticFrames
        self.bs, self.hw, self.device, self.step = bs, input_hw, device, 0
        self.raw_hw = raw_hw
        rng = np.random.RandomState(seed)
        self.conf = dict(resize_lim=(0.40, 0.47), final_dim=tuple(input_hw), bot_pct_lim=(0.0, 0.0), rot_lim=(-5.4, 5.4),
                         H=raw_hw[0], W=raw_hw[1], rand_flip=True, rot3d_range=[0, 0])
        self.dataset = _ToySequences(max(num_seq, bs), seq_len, self.conf, rng)
        self.sampler = iter(GroupInBatchSampler(self.dataset, batch_size=bs, world_size=1, rank=0, seed=seed))
        g = torch.Generator().manual_seed(seed)
        self.raw = [torch.randint(0, 256, (6,) + tuple(raw_hw) + (3,), generator=g, dtype=torch.uint8).to(device)
                    for _ in range(pool)]
        self.lidar2img = syn.bench2drive_lidar2img()                  # (6, 4, 4) float64, camera resolution
        self.mean, self.std = np.asarray(mean, np.float32), np.asarray(std, np.float32)
        self._gt = SyntheticFrames(bs=bs, input_hw=input_hw, device=device, seed=seed)   # ground truth + command + target
        self.last_aug = None

    def next(self):
        from . import imgpipe
        batch = next(self.sampler)
        imgs, mats = [], []
        for item in batch:
            aug = item["aug_config"]
            raw = self.raw[item["idx"] % len(self.raw)]
            # two launches: horizontal resample of the rows the crop needs; vertical resample + crop + flip + rotate +
            # BGR->RGB + normalise + CHW store
            imgs.append(imgpipe.transform_images(raw, aug, self.mean, self.std, True, layout="chw"))
            mats.append(imgpipe.transform_matrix(aug, *self.raw_hw) @ self.lidar2img)      # (4,4) @ (6,4,4): one product
        self.last_aug = [item["aug_config"] for item in batch]
        _, data = self._gt.next()
        img = torch.stack(imgs, 0)                                                          # (bs, 6, 3, h, w) float32
        data = dict(data)
        data["projection_mat"] = torch.from_numpy(np.stack(mats).astype(np.float32)).to(self.device, non_blocking=True)
        # every frame of a sequence moves the ego pose on (planar motion, as SyntheticFrames); a new sequence starts 1000 s
        # later, which is what invalidates the temporal caches (instance banks: |dt| <= max_time_interval)
        k = self.step
        L = self.dataset.seq_len
        T = [syn.ego_motion(item["idx"] % L) for item in batch]
        data["img_metas"] = [dict(T_global=t, T_global_inv=np.linalg.inv(t)) for t in T]
        data["timestamp_host"] = torch.tensor([0.5 * (item["idx"] % L) + 1e3 * (item["idx"] // L) for item in batch],
                                              dtype=torch.float64)
        data["timestamp"] = data["timestamp_host"].to(self.device)
        self.step = k + 1
        return img, data
