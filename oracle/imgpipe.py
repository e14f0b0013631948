"""CPU restatement of the image leg of the reference's training pipeline -- TEST INFRASTRUCTURE ONLY.

Reference: projects/mmdet3d_plugin/datasets/pipelines/augment.py:11-94 (ResizeCropFlipImage._img_transform:
PIL ``resize`` -> ``crop`` -> ``transpose(FLIP_LEFT_RIGHT)`` -> ``rotate``, then float32) and
pipelines/transform.py:286-321 (NormalizeMultiviewImage -> mmcv.imnormalize) and :136-138 (HWC -> CHW).

The pixel arithmetic lives in a third-party dependency, Pillow (importable here: 12.2.0; the reference pins none).
Its published algorithm, restated below in numpy integer arithmetic:
  * Image.resize, default filter BICUBIC (a = -0.5, support 2): two separable passes, horizontal first, uint8 between
    them; per output index the taps are [xmin, xmin + n) with xmin = int(center - support + 0.5) clamped,
    center = (i + 0.5) * scale, support = 2 * max(scale, 1); weights = filter((x + xmin - center + 0.5) / max(scale, 1))
    normalised to sum 1 in double, then rounded half away from zero to 22 fractional bits; a pixel is
    clip8((2^21 + sum w_k p_k) >> 22) (src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc,
    ImagingResampleHorizontal_8bpc / Vertical_8bpc).  A pass whose size does not change is skipped.
  * Image.crop: integer box, zero fill outside the image.
  * Image.rotate(angle): nearest neighbour about (w / 2, h / 2) in 16.16 fixed point, zero fill
    (PIL/Image.py rotate: matrix rounded to 15 decimals; src/libImaging/Geometry.c affine_fixed).
Pinned by tests/test_imgpipe_cpu.py against Pillow itself (bit-exact) and against the reference's own _img_transform
(tests/golden/image_pipeline.npz).  mmcv.imnormalize (cv2) is absent here: its arithmetic -- float32 (x - mean) *
float32(1 / float64(std)) after the BGR -> RGB swap -- is restated from mmcv==1.7.1 image/photometric.py:imnormalize_
and stays "parity unpinned".
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_tables(in_size, out_size):
    """-> (ksize, bounds (out_size, 2) int32 [first, count], coeffs (out_size, ksize) int32)."""
    if in_size == out_size:          # the pass is skipped: identity taps
        bounds = np.stack([np.arange(out_size), np.ones(out_size, np.int64)], 1).astype(np.int32)
        return 1, bounds, np.full((out_size, 1), 1 << PRECISION_BITS, np.int32)
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    coeffs = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            coeffs[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, coeffs


def _pass(img, bounds, coeffs, axis):
    """One separable pass over ``axis`` (0 = rows / vertical, 1 = columns / horizontal) of a (H, W, C) uint8 image."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.uint8)
    for i, (first, n) in enumerate(bounds):
        acc = np.tensordot(coeffs[i, :n].astype(np.int64), src[first:first + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[i] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis)


def resize(img, out_w, out_h):
    """PIL Image.resize((out_w, out_h)) of a (H, W, C) uint8 array."""
    h, w = img.shape[:2]
    _, bh, ch = resample_tables(w, out_w)
    _, bv, cv = resample_tables(h, out_h)
    if w != out_w:
        # only the rows the vertical pass reads are resampled (same values; the restatement keeps the whole image)
        img = _pass(img, bh, ch, 1)
    if h != out_h:
        img = _pass(img, bv, cv, 0)
    return img.copy()


def crop(img, box):
    x0, y0, x1, y1 = box
    out = np.zeros((y1 - y0, x1 - x0) + img.shape[2:], img.dtype)
    h, w = img.shape[:2]
    sx0, sy0, sx1, sy1 = max(x0, 0), max(y0, 0), min(x1, w), min(y1, h)
    if sx1 > sx0 and sy1 > sy0:
        out[sy0 - y0:sy1 - y0, sx0 - x0:sx1 - x0] = img[sy0:sy1, sx0:sx1]
    return out


def rotate_fixed(angle, w, h):
    """Image.rotate(angle)'s inverse map in 16.16 fixed point: (a0, a1, a2, a3, a4, a5), or None for the copy path."""
    angle = angle % 360.0
    if angle == 0:
        return None
    if angle == 180 or (angle in (90, 270) and w == h):
        raise NotImplementedError("PIL takes its transpose fast path for %r; not needed by the pipeline" % angle)
    cx, cy = w / 2.0, h / 2.0
    r = -math.radians(angle)
    m = [round(math.cos(r), 15), round(math.sin(r), 15), 0.0, round(-math.sin(r), 15), round(math.cos(r), 15), 0.0]
    m[2] = m[0] * -cx + m[1] * -cy + m[2]
    m[5] = m[3] * -cx + m[4] * -cy + m[5]
    m[2] += cx
    m[5] += cy

    def fix(v):
        v = v * 65536.0 + 0.5
        return int(math.floor(v)) if v < 0.0 else int(v)

    return (fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]), fix(m[5] + m[3] * 0.5 + m[4] * 0.5))


def rotate(img, angle):
    h, w = img.shape[:2]
    a = rotate_fixed(angle, w, h)
    if a is None:
        return img.copy()
    a0, a1, a2, a3, a4, a5 = a
    ys, xs = np.mgrid[0:h, 0:w].astype(np.int64)
    xin = (a2 + a1 * ys + a0 * xs) >> 16
    yin = (a5 + a4 * ys + a3 * xs) >> 16
    ok = (xin >= 0) & (xin < w) & (yin >= 0) & (yin < h)
    out = np.zeros_like(img)
    out[ok] = img[yin[ok], xin[ok]]
    return out


def img_transform(img, resize_factor, crop_box, flip, angle):
    """ResizeCropFlipImage._img_transform on a uint8 (H, W, 3) image -> float32 (h, w, 3) (augment.py:46-68)."""
    H, W = img.shape[:2]
    out = resize(img, int(W * resize_factor), int(H * resize_factor))
    out = crop(out, crop_box)
    if flip:
        out = out[:, ::-1]
    out = rotate(np.ascontiguousarray(out), angle)
    return out.astype(np.float32)


def transform_matrix(resize_factor, crop_box, flip, angle):
    """The 4x4 pixel-space matrix the reference multiplies onto lidar2img (augment.py:70-94), float64."""
    m = np.eye(3)
    m[:2, :2] *= resize_factor
    m[:2, 2] -= np.array(crop_box[:2])
    if flip:
        m = np.array([[-1, 0, crop_box[2] - crop_box[0]], [0, 1, 0], [0, 0, 1]]) @ m
    r = angle / 180 * np.pi
    rot = np.array([[np.cos(r), np.sin(r), 0], [-np.sin(r), np.cos(r), 0], [0, 0, 1]])
    center = np.array([crop_box[2] - crop_box[0], crop_box[3] - crop_box[1]]) / 2
    rot[:2, 2] = -rot[:2, :2] @ center + center
    m = rot @ m
    ext = np.eye(4)
    ext[:3, :3] = m
    return ext


def imnormalize(img, mean, std, to_rgb=True):
    """mmcv.imnormalize on a float32 (h, w, 3) image (restated; see the module docstring)."""
    img = img.astype(np.float32)
    if to_rgb:
        img = img[..., ::-1]
    mean = np.asarray(mean, np.float32).reshape(1, 1, 3)
    stdinv = (1.0 / np.asarray(std, np.float32).astype(np.float64)).astype(np.float32).reshape(1, 1, 3)
    return (img - mean) * stdinv
