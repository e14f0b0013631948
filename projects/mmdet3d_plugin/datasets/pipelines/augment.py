"""``ResizeCropFlipImage`` (reference datasets/pipelines/augment.py:11-94) on device tensors.

``results["img"]`` is the uint8 (n, H, W, 3) device tensor of one sample's camera frames (or a list of n (H, W, 3)
device tensors of one size); the n PIL round trips of the reference become two launches for the whole sample
(hipad_amd.imgpipe.transform_images, geometry bit-exact with Pillow).  The projection matrices are composed on the host
in float64 (stacked 4x4 products, same values as the reference's per-camera loop).  The BEV rotation augmentation
(reference augment.py:95-138, off in the stage configs: rot3d_range = [0, 0]) lives in hipad_amd.dataflow.rotate_scene."""
import numpy as np
import torch

from hipad_amd import imgpipe
from hipad_amd.compat import PIPELINES

__all__ = ["ResizeCropFlipImage", "compose_camera_matrices"]


def compose_camera_matrices(results, image_matrix, resize):
    """The image-space matrix of the augmentation applied to every camera's projection at once (stacked 4x4 products;
    the reference multiplies camera by camera, augment.py:24-29): lidar2img / ego2img pick it up on the left, the
    intrinsics scale with the resize factor."""
    for key in ("lidar2img", "ego2img"):
        if key in results:
            results[key] = list(image_matrix @ np.stack(results[key]))
    if "cam_intrinsic" in results:
        k = np.stack(results["cam_intrinsic"])
        k[:, :3, :3] *= resize
        results["cam_intrinsic"] = list(k)


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
        compose_camera_matrices(results, imgpipe.transform_matrix(aug_config, H, W), aug_config["resize"])
        results["img"] = list(out.unbind(0))
        results["img_shape"] = [tuple(x.shape[:2]) for x in results["img"]]
        return results
