"""ctypes loader for oracle/daf_oracle.c (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

numpy in, numpy out.  Follows the reference's operator contract
(projects/mmdet3d_plugin/ops/src/deformable_aggregation.cpp:23-29, 31-62, 86-124).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libhipad_oracle.so")
_lib = None


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    src = os.path.join(_HERE, "daf_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        for name in ("hipad_oracle_daf_forward", "hipad_oracle_daf_backward", "hipad_oracle_daf_taps"):
            getattr(_lib, name).restype = ctypes.c_int
        _lib.hipad_oracle_set_threads(1)   # the checker walks the reference's index space sequentially
    return _lib


def set_threads(n):
    """Threads of the C restatement (1 = the sequential checker; >1 only for bench.py's cpu_baseline timing: anchors are
    dealt to threads and grad_feat is scattered with atomic adds, like the reference CUDA kernel)."""
    lib().hipad_oracle_set_threads(int(n))


def _f32(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def _i32(x):
    return np.ascontiguousarray(x, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _dims(feat, spatial_shape, loc, weights):
    bs, num_feat, C = feat.shape
    cams, scales = spatial_shape.shape[:2]
    A, P = loc.shape[1:3]
    G = weights.shape[5]
    assert loc.shape == (bs, A, P, cams, 2), loc.shape
    assert weights.shape == (bs, A, P, cams, scales, G), weights.shape
    return bs, cams, num_feat, C, scales, A, P, G


def daf_forward(feat, spatial_shape, scale_start_index, loc, weights, acc64=False):
    feat, loc, weights = _f32(feat), _f32(loc), _f32(weights)
    ss, st = _i32(spatial_shape), _i32(scale_start_index)
    bs, cams, num_feat, C, scales, A, P, G = _dims(feat, ss, loc, weights)
    out = np.zeros((bs, A, C), np.float32)
    rc = lib().hipad_oracle_daf_forward(
        _p(feat), _p(ss), _p(st), _p(loc), _p(weights), _p(out),
        bs, cams, num_feat, C, scales, A, P, G, int(acc64))
    if rc:
        raise RuntimeError(f"oracle forward rc={rc}")
    return out


def daf_backward(feat, spatial_shape, scale_start_index, loc, weights, grad_out, acc64=False):
    feat, loc, weights, grad_out = _f32(feat), _f32(loc), _f32(weights), _f32(grad_out)
    ss, st = _i32(spatial_shape), _i32(scale_start_index)
    bs, cams, num_feat, C, scales, A, P, G = _dims(feat, ss, loc, weights)
    assert grad_out.shape == (bs, A, C)
    gf = np.zeros_like(feat)
    gl = np.zeros_like(loc)
    gw = np.zeros_like(weights)
    rc = lib().hipad_oracle_daf_backward(
        _p(feat), _p(ss), _p(st), _p(loc), _p(weights), _p(grad_out), _p(gf), _p(gl), _p(gw),
        bs, cams, num_feat, C, scales, A, P, G, int(acc64))
    if rc:
        raise RuntimeError(f"oracle backward rc={rc}")
    return gf, gl, gw


def daf_taps(spatial_shape, scale_start_index, loc, num_feat):
    """Index work only: (valid u8 [bs,A,P,cams], taps i32 [bs,A,P,cams,scales,4])."""
    loc = _f32(loc)
    ss, st = _i32(spatial_shape), _i32(scale_start_index)
    bs, A, P, cams = loc.shape[:4]
    scales = ss.shape[1]
    valid = np.zeros((bs, A, P, cams), np.uint8)
    taps = np.zeros((bs, A, P, cams, scales, 4), np.int32)
    lib().hipad_oracle_daf_taps(_p(ss), _p(st), _p(loc), _p(valid), _p(taps),
                                bs, cams, num_feat, scales, A, P)
    return valid, taps
