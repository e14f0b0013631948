"""``NormalizeMultiviewImage`` (reference datasets/pipelines/transform.py:286-321) on device tensors, and
``DeviceImageTransform`` (ours): ResizeCropFlipImage + NormalizeMultiviewImage + the HWC->CHW stack of
NuScenesSparse4DAdaptor (transform.py:136-138) and its projection_mat / image_wh entries (:112-116) as ONE pipeline
step = two kernel launches per sample, for pipelines with nothing between the resize and the normalisation."""
import numpy as np
import torch

from hipad_amd import imgpipe
from hipad_amd.compat import PIPELINES

__all__ = ["NormalizeMultiviewImage", "DeviceImageTransform", "NuScenesSparse4DAdaptor"]


@PIPELINES.register_module()
class NormalizeMultiviewImage(object):
    """(x[BGR->RGB] - mean) * (1 / std) in float32, mmcv.imnormalize's arithmetic, on (h, w, 3) device tensors."""

    def __init__(self, mean, std, to_rgb=True):
        self.mean = np.array(mean, dtype=np.float32)
        self.std = np.array(std, dtype=np.float32)
        self.to_rgb = to_rgb

    def __call__(self, results):
        imgs = results["img"]
        x = imgs if isinstance(imgs, torch.Tensor) else torch.stack(list(imgs), 0)
        x = x.float()
        if self.to_rgb:
            x = x.flip(-1)
        mean = torch.from_numpy(self.mean).to(x.device)
        stdinv = torch.from_numpy((1.0 / self.std.astype(np.float64)).astype(np.float32)).to(x.device)
        x = (x - mean) * stdinv
        results["img"] = list(x.unbind(0))
        results["img_norm_cfg"] = dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb)
        return results

    def __repr__(self):
        return f"{self.__class__.__name__}(mean={self.mean}, std={self.std}, to_rgb={self.to_rgb})"


@PIPELINES.register_module()
class DeviceImageTransform(object):
    def __init__(self, mean, std, to_rgb=True, channels_last=False):
        self.mean = np.array(mean, dtype=np.float32)
        self.std = np.array(std, dtype=np.float32)
        self.to_rgb = to_rgb
        self.channels_last = channels_last

    def __call__(self, results):
        src = results["img"] if isinstance(results["img"], torch.Tensor) else torch.stack(list(results["img"]), 0)
        aug_config = results.get("aug_config") or {}
        n, H, W = src.shape[:3]
        img = imgpipe.transform_images(src.contiguous(), aug_config, self.mean, self.std, self.to_rgb, layout="chw",
                                       channels_last=self.channels_last)
        if results.get("aug_config") is not None:
            mat = imgpipe.transform_matrix(aug_config, H, W)
            for i in range(n):
                results["lidar2img"][i] = mat @ results["lidar2img"][i]
                if "ego2img" in results:
                    results["ego2img"][i] = mat @ results["ego2img"][i]
                if "cam_intrinsic" in results:
                    results["cam_intrinsic"][i][:3, :3] *= aug_config["resize"]
        h, w = img.shape[-2:]
        results["img"] = img                                            # (n, 3, h, w) float32: what the adaptor stacks
        results["img_shape"] = [(h, w)] * n
        results["img_norm_cfg"] = dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb)
        if "lidar2img" in results:
            results["projection_mat"] = np.float32(np.stack(results["lidar2img"]))
            results["image_wh"] = np.ascontiguousarray(np.array(results["img_shape"], dtype=np.float32)[:, :2][:, ::-1])
        return results


def _tensor(x):
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x))


@PIPELINES.register_module()
class NuScenesSparse4DAdaptor(object):
    """Last pipeline step before ``Collect`` (reference datasets/pipelines/transform.py:107-168): the per-sample
    projection matrices / image sizes / pose inverses the decoder reads, yaw wrapped into (-pi, pi], ground truth as
    tensors, images stacked channels-first.  Host logic; the values are plain tensors (mmcv's DataContainer only tells
    its collate function how to batch them -- hipad_amd.frame batches by stacking / padding itself).  Images that are
    already a (n, 3, h, w) tensor (DeviceImageTransform) or device (h, w, 3) tensors stay on their device."""

    GT_LIST_KEYS = ("gt_map_labels", "gt_map_pts", "gt_agent_fut_trajs", "gt_agent_fut_masks")
    GT_STACK_KEYS = ("gt_ego_fut_trajs", "gt_ego_fut_masks", "gt_ego_fut_cmd", "command_near_xy", "ego_status")

    def __call__(self, input_dict):
        input_dict["projection_mat"] = np.float32(np.stack(input_dict["lidar2img"]))
        input_dict["image_wh"] = np.ascontiguousarray(np.array(input_dict["img_shape"], dtype=np.float32)[:, :2][:, ::-1])
        input_dict["T_global_inv"] = np.linalg.inv(input_dict["lidar2global"])
        input_dict["T_global"] = input_dict["lidar2global"]
        if "cam_intrinsic" in input_dict:
            input_dict["cam_intrinsic"] = np.float32(np.stack(input_dict["cam_intrinsic"]))
            input_dict["focal"] = input_dict["cam_intrinsic"][..., 0, 0]
        if "instance_inds" in input_dict:
            input_dict["instance_id"] = input_dict["instance_inds"]
        if "gt_bboxes_3d" in input_dict:
            boxes = input_dict["gt_bboxes_3d"]
            boxes[:, 6] = self.limit_period(boxes[:, 6], offset=0.5, period=2 * np.pi)
            input_dict["gt_bboxes_3d"] = _tensor(boxes).float()
        if "gt_labels_3d" in input_dict:
            input_dict["gt_labels_3d"] = _tensor(input_dict["gt_labels_3d"]).long()
        img = input_dict["img"]
        if isinstance(img, torch.Tensor) and img.dim() == 4:
            input_dict["img"] = img                                           # already (n, 3, h, w)
        elif isinstance(img[0], torch.Tensor):
            input_dict["img"] = torch.stack(list(img), 0).permute(0, 3, 1, 2).contiguous()
        else:
            input_dict["img"] = _tensor(np.ascontiguousarray(np.stack([im.transpose(2, 0, 1) for im in img], axis=0)))
        for key in self.GT_LIST_KEYS + self.GT_STACK_KEYS:
            if key in input_dict:
                input_dict[key] = _tensor(input_dict[key])
        return input_dict

    @staticmethod
    def limit_period(val, offset=0.5, period=np.pi):
        return val - np.floor(val / period + offset) * period
