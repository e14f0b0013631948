"""Host side of the device image pipeline (csrc/imgpipe.hip; include/hipad.h "Image leg of the training data pipeline").

The raw camera frames of one sample sit in HBM as uint8 (n, H, W, 3) in file channel order (BGR, as mmcv.imread gives
them); ``transform_images`` turns them into the float32 tensor the detector consumes with two kernel launches, the
geometry bit-exact with the PIL calls of the reference's ResizeCropFlipImage (datasets/pipelines/augment.py:46-68).
The tap tables come from the library's own host functions and are cached on the device per (source size, aug_config):
with ``keep_consistent_seq_aug`` (the stage-2 config) a whole driving sequence shares one entry.
"""
import collections
import ctypes

import numpy as np
import torch

from . import lib as _lib


def resample_tables(in_size, out_size):
    """-> (ksize, bounds (out_size, 2) int32, coeffs (out_size, ksize) int32) as Pillow's bicubic resize builds them."""
    lib = _lib.load()
    ksize = lib.hipad_resample_tables(int(in_size), int(out_size), None, None)
    if ksize <= 0:
        _lib.check(ksize, "hipad_resample_tables")
    bounds = np.empty((out_size, 2), np.int32)
    coeffs = np.empty((out_size, ksize), np.int32)
    got = lib.hipad_resample_tables(int(in_size), int(out_size), bounds.ctypes.data, coeffs.ctypes.data)
    assert got == ksize
    return ksize, bounds, coeffs


def rotate_fixed(angle, width, height):
    """Image.rotate(angle)'s 16.16 inverse map (a0 .. a5) or None when the rotation is the identity."""
    a = np.zeros(6, np.int32)
    r = _lib.load().hipad_rotate_fixed(float(angle), int(width), int(height), a.ctypes.data)
    if r < 0:
        _lib.check(r, "hipad_rotate_fixed")
    return tuple(int(v) for v in a) if r == 1 else None


Plan = collections.namedtuple("Plan", "res_w res_h row0 rows ksize_h ksize_v bounds_h coeffs_h bounds_v coeffs_v crop flip rot")


def make_plan(src_h, src_w, aug_config, device=None):
    """Everything ``transform_images`` needs for one (source size, aug_config); tables on ``device`` when given."""
    resize = aug_config.get("resize", 1)
    res_w, res_h = int(src_w * resize), int(src_h * resize)          # augment.py:48-49
    crop = tuple(int(v) for v in aug_config.get("crop", [0, 0, res_w, res_h]))
    flip = bool(aug_config.get("flip", False))
    rot = rotate_fixed(aug_config.get("rotate", 0), crop[2] - crop[0], crop[3] - crop[1])
    ksize_h, bounds_h, coeffs_h = resample_tables(src_w, res_w)
    ksize_v, bounds_v, coeffs_v = resample_tables(src_h, res_h)
    # the horizontal pass only produces the source rows the vertical pass reads (Pillow does the same: ybox_first/last)
    row0 = int(bounds_v[0, 0])
    rows = int(bounds_v[-1, 0] + bounds_v[-1, 1]) - row0
    bounds_v = bounds_v.copy()
    bounds_v[:, 0] -= row0

    def up(a):
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t.to(device) if device is not None else t

    return Plan(res_w, res_h, row0, rows, ksize_h, ksize_v, up(bounds_h), up(coeffs_h), up(bounds_v), up(coeffs_v), crop, flip, rot)


_plans = collections.OrderedDict()


def _cached_plan(src_h, src_w, aug_config, device):
    key = (src_h, src_w, float(aug_config.get("resize", 1)), tuple(aug_config.get("crop", ())), bool(aug_config.get("flip", False)),
           float(aug_config.get("rotate", 0)), str(device))
    plan = _plans.get(key)
    if plan is None:
        plan = _plans[key] = make_plan(src_h, src_w, aug_config, device)
        while len(_plans) > 64:
            _plans.popitem(last=False)
    return plan


def transform_images(src, aug_config, mean=None, std=None, to_rgb=True, layout="chw", channels_last=False):
    """uint8 (n, H, W, 3) device tensor -> float32 images of the crop's size.

    ``layout`` "chw": (n, 3, h, w) (the tensor NuScenesSparse4DAdaptor stacks; ``channels_last`` stores it in
    torch.channels_last strides for the encoder), "hwc": (n, h, w, 3) (what ResizeCropFlipImage alone returns).
    ``mean`` / ``std`` None: raw pixel values, no channel swap unless ``to_rgb``."""
    lib = _lib.load()
    _lib._req(src, torch.uint8, "src")
    if src.dim() != 4 or src.shape[-1] != 3:
        raise _lib.HipadError("src must be (n, H, W, 3) uint8")
    n, H, W, _ = src.shape
    plan = _cached_plan(H, W, aug_config, src.device)
    out_w, out_h = plan.crop[2] - plan.crop[0], plan.crop[3] - plan.crop[1]
    tmp = torch.empty(n, plan.rows, plan.res_w, 3, dtype=torch.uint8, device=src.device)
    if layout == "chw":
        out = torch.empty(n, 3, out_h, out_w, dtype=torch.float32, device=src.device,
                          memory_format=torch.channels_last if channels_last else torch.contiguous_format)
        strides = (out.stride(0), out.stride(1), out.stride(2), out.stride(3))
    elif layout == "hwc":
        out = torch.empty(n, out_h, out_w, 3, dtype=torch.float32, device=src.device)
        strides = (out.stride(0), out.stride(3), out.stride(1), out.stride(2))
    else:
        raise ValueError(layout)
    c_strides = (ctypes.c_longlong * 4)(*strides)
    c_crop = (ctypes.c_int * 4)(*plan.crop)
    c_rot = (ctypes.c_int * 6)(*plan.rot) if plan.rot is not None else None
    c_mean = c_std = None
    if mean is not None:
        c_mean = (ctypes.c_float * 3)(*[float(v) for v in mean])
        c_std = (ctypes.c_float * 3)(*[float(v) for v in std])
    with torch.cuda.device(src.device):
        st = _lib.stream_ptr(src.device)
        _lib.check(lib.hipad_image_resize_rows(tmp.data_ptr(), src.data_ptr(), plan.bounds_h.data_ptr(), plan.coeffs_h.data_ptr(),
                                               plan.ksize_h, n, H, W, plan.row0, plan.rows, plan.res_w, st),
                   "hipad_image_resize_rows")
        _lib.check(lib.hipad_image_finish(out.data_ptr(), c_strides, tmp.data_ptr(), plan.bounds_v.data_ptr(),
                                          plan.coeffs_v.data_ptr(), plan.ksize_v, n, plan.rows, plan.res_w, plan.res_h, c_crop,
                                          int(plan.flip), c_rot, c_mean, c_std, int(bool(to_rgb)), st), "hipad_image_finish")
    return out


def transform_matrix(aug_config, src_h, src_w):
    """4x4 float64 pixel-space matrix of the same augmentation (reference augment.py:70-94): new lidar2img = M @ lidar2img."""
    resize = aug_config.get("resize", 1)
    crop = aug_config.get("crop", [0, 0, int(src_w * resize), int(src_h * resize)])
    m = np.eye(3)
    m[:2, :2] *= resize
    m[:2, 2] -= np.array(crop[:2])
    if aug_config.get("flip", False):
        m = np.array([[-1, 0, crop[2] - crop[0]], [0, 1, 0], [0, 0, 1]]) @ m
    r = aug_config.get("rotate", 0) / 180 * np.pi
    rot = np.array([[np.cos(r), np.sin(r), 0], [-np.sin(r), np.cos(r), 0], [0, 0, 1]])
    center = np.array([crop[2] - crop[0], crop[3] - crop[1]]) / 2
    rot[:2, 2] = -rot[:2, :2] @ center + center
    ext = np.eye(4)
    ext[:3, :3] = rot @ m
    return ext
