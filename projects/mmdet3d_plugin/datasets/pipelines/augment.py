"""``ResizeCropFlipImage`` (reference datasets/pipelines/augment.py:11-94) on device tensors.

``results["img"]`` is the uint8 (n, H, W, 3) device tensor of one sample's camera frames (or a list of n (H, W, 3)
device tensors of one size); the n PIL round trips of the reference become two launches for the whole sample
(hipad_amd.imgpipe.transform_images, geometry bit-exact with Pillow).  The projection matrices are composed on the host
in float64 exactly as the reference does (4x4 matrices, n of them)."""
import numpy as np
import torch

from hipad_amd import imgpipe
from hipad_amd.compat import PIPELINES

__all__ = ["ResizeCropFlipImage", "BBoxRotation"]


def _stack(imgs):
    if isinstance(imgs, torch.Tensor):
        return imgs
    return torch.stack(list(imgs), 0)


@PIPELINES.register_module()
class ResizeCropFlipImage(object):
    def __init__(self, with_img_depth=False):
        if with_img_depth:
            raise NotImplementedError("with_img_depth: the stage configs of this path do not use image depth maps")
        self.with_img_depth = with_img_depth

    def __call__(self, results):
        aug_config = results.get("aug_config")
        if aug_config is None:
            return results
        src = _stack(results["img"])
        if src.dtype != torch.uint8:
            raise TypeError("device ResizeCropFlipImage takes the uint8 frames as loaded (got %s)" % src.dtype)
        n, H, W = src.shape[:3]
        out = imgpipe.transform_images(src.contiguous(), aug_config, layout="hwc", to_rgb=False)
        mat = imgpipe.transform_matrix(aug_config, H, W)
        for i in range(n):
            results["lidar2img"][i] = mat @ results["lidar2img"][i]
            if "ego2img" in results:
                results["ego2img"][i] = mat @ results["ego2img"][i]
            if "cam_intrinsic" in results:
                results["cam_intrinsic"][i][:3, :3] *= aug_config["resize"]
        results["img"] = list(out.unbind(0))
        results["img_shape"] = [tuple(x.shape[:2]) for x in results["img"]]
        return results


@PIPELINES.register_module()
class BBoxRotation(object):
    """Rotation of the scene about the vertical axis by ``aug_config["rotate_3d"]`` (reference
    datasets/pipelines/augment.py:95-138): the lidar -> image and lidar -> global matrices absorb the inverse rotation,
    box centres / yaws / velocities turn with the scene.  Host logic on 4x4 matrices and a handful of boxes (numpy)."""

    def __call__(self, results):
        angle = results["aug_config"]["rotate_3d"]
        c, s_ = np.cos(angle), np.sin(angle)
        undo = np.linalg.inv(np.array([[c, -s_, 0, 0], [s_, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]]))
        results["lidar2img"] = [m @ undo for m in results["lidar2img"]]
        if "lidar2global" in results:
            results["lidar2global"] = results["lidar2global"] @ undo
        if "gt_bboxes_3d" in results:
            results["gt_bboxes_3d"] = self.box_rotate(results["gt_bboxes_3d"], angle)
        return results

    @staticmethod
    def box_rotate(bbox_3d, angle):
        c, s_ = np.cos(angle), np.sin(angle)
        turn = np.array([[c, s_, 0], [-s_, c, 0], [0, 0, 1]])           # row vectors: p' = p @ turn
        bbox_3d[:, :3] = bbox_3d[:, :3] @ turn
        bbox_3d[:, 6] += angle
        if bbox_3d.shape[-1] > 7:
            n = bbox_3d[:, 7:].shape[-1]
            bbox_3d[:, 7:] = bbox_3d[:, 7:] @ turn[:n, :n]
        return bbox_3d
