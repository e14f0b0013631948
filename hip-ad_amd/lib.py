"""ctypes binding of libhipad.so (the C ABI declared in include/hipad.h).

The product path fails loudly when the library is missing or a symbol is absent: there is
no CPU / eager fallback anywhere behind these functions.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("HIPAD_LIB") or os.path.join(_HERE, "csrc", "libhipad.so")  # HIPAD_LIB: another build of the same ABI (A/B timing)

c_int, c_void_p, c_size_t = ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/hipad.h declares
SIGNATURES = {
    "hipad_abi_version": (c_int, []),
    "hipad_status_string": (ctypes.c_char_p, [c_int]),
    "hipad_daf_set_pairs_per_wave": (None, [c_int, c_int]),
    "hipad_daf_forward_workspace": (c_size_t, [c_int] * 8),
    "hipad_daf_forward": (c_int, [c_void_p] * 6 + [c_int] * 8 + [c_void_p, c_size_t, c_void_p]),
    "hipad_daf_forward_bf16": (c_int, [c_void_p] * 6 + [c_int] * 8 + [c_void_p, c_size_t, c_void_p]),
    "hipad_daf_backward_bf16": (c_int, [c_void_p] * 9 + [c_int] * 8 + [c_int, c_void_p, c_size_t, c_void_p]),
    "hipad_daf_backward_workspace": (c_size_t, [c_int] * 8),
    "hipad_daf_backward": (c_int, [c_void_p] * 9 + [c_int] * 8 + [c_int, c_void_p, c_size_t, c_void_p]),
    "hipad_daf_taps": (c_int, [c_void_p] * 5 + [c_int] * 6 + [c_void_p]),
    "hipad_daf_set_tap_chunks": (None, [c_int]),
    "hipad_daf_set_feat_run": (None, [c_int]),
    "hipad_daf_set_feat_blocks": (None, [c_int]),
    "hipad_weights_softmax_set_split": (None, [c_int]),
    "hipad_daf_backward_feat_multi_workspace": (c_size_t, [c_void_p, c_int] + [c_int] * 6),
    "hipad_daf_backward_feat_multi": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 6
                                      + [c_void_p, c_size_t, c_void_p]),
    "hipad_project_points_forward": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "hipad_project_points_backward": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p]),
    "hipad_weights_softmax_forward": (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_void_p]),
    "hipad_weights_softmax_backward_workspace": (c_size_t, [c_int] * 7),
    "hipad_weights_softmax_backward": (c_int, [c_void_p] * 7 + [c_int] * 7 + [c_void_p, c_size_t, c_void_p]),
    "hipad_linear_forward": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "hipad_linear_backward": (c_int, [c_void_p] * 7 + [c_int] * 3 + [c_void_p]),
    "hipad_box_points_project_forward": (c_int, [c_void_p] * 7 + [c_int] * 6 + [c_void_p]),
    "hipad_box_points_project_backward": (c_int, [c_void_p] * 8 + [c_int] * 6 + [c_void_p]),
    "hipad_line_points_project_forward": (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_void_p]),
    "hipad_line_points_project_backward": (c_int, [c_void_p] * 8 + [c_int] * 6 + [c_void_p]),
    "hipad_linear_relu_ln_supported": (c_int, [c_int, c_int]),
    "hipad_linear_relu_ln_forward": (c_int, [c_void_p] * 9 + [c_int] * 3 + [ctypes.c_float, c_void_p]),
    "hipad_layernorm_forward": (c_int, [c_void_p] * 6 + [c_int, c_int, ctypes.c_float, c_void_p]),
    "hipad_layernorm_backward": (c_int, [c_void_p] * 8 + [c_int, c_int, c_void_p]),
    "hipad_linear_assignment": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p]),
    "hipad_focal_loss_forward": (c_int, [c_void_p] * 6 + [ctypes.c_longlong, c_int, c_int, ctypes.c_float, ctypes.c_float, c_void_p]),
    "hipad_loss_det_assign": (c_int, [c_void_p] * 11 + [c_int] * 7 + [c_void_p]),
    "hipad_loss_det": (c_int, [c_void_p] * 13 + [c_int] * 8 + [c_void_p]),
    "hipad_loss_map_assign": (c_int, [c_void_p] * 13 + [c_int] * 7 + [c_void_p]),
    "hipad_loss_map": (c_int, [c_void_p] * 11 + [c_int] * 7 + [c_void_p]),
    "hipad_loss_motion": (c_int, [c_void_p] * 7 + [c_int] + [c_void_p] * 3 + [c_int] * 6 + [c_void_p]),
    "hipad_loss_plan": (c_int, [c_void_p] * 8 + [c_int] * 6 + [c_void_p]),
    "hipad_loss_scale": (c_int, [c_void_p] * 5 + [c_int, c_void_p]),
    "hipad_adamw_workspace": (c_size_t, []),
    "hipad_adamw_step": (c_int, [c_void_p] * 4 + [ctypes.c_longlong] * 2 + [ctypes.c_float] * 7
                         + [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_void_p, c_void_p]),
    "hipad_lr_factor": (ctypes.c_float, [c_void_p, c_int]),
    "hipad_keep_mask": (c_int, [c_void_p, ctypes.c_longlong, ctypes.c_float, ctypes.c_uint, c_void_p, c_void_p]),
    "hipad_chunk_mix": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    "hipad_bn_supported": (c_int, [ctypes.c_longlong, c_int]),
    "hipad_bn_forward": (c_int, [c_void_p] * 9 + [ctypes.c_longlong, c_int, ctypes.c_float, ctypes.c_float, c_int, c_void_p]),
    "hipad_bn_forward_grouped": (c_int, [c_void_p] * 9 + [ctypes.c_longlong, c_int, ctypes.c_float, ctypes.c_float, c_int,
                                         ctypes.c_longlong, ctypes.c_longlong, c_void_p]),
    "hipad_bn_backward": (c_int, [c_void_p] * 10 + [ctypes.c_longlong, c_int, c_void_p]),
    "hipad_grid_mask": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    "hipad_dropout_add": (c_int, [c_void_p] * 3 + [ctypes.c_longlong, ctypes.c_float, ctypes.c_uint, c_void_p, c_void_p]),
    "hipad_resample_tables": (c_int, [c_int, c_int, c_void_p, c_void_p]),
    "hipad_rotate_fixed": (c_int, [ctypes.c_double, c_int, c_int, c_void_p]),
    "hipad_image_resize_rows": (c_int, [c_void_p] * 4 + [c_int] * 7 + [c_void_p]),
    "hipad_image_finish": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                                                      c_void_p]),
    "hipad_motion_query_embed": (c_int, [c_void_p] * 5 + [ctypes.c_longlong] + [c_int] * 7 + [c_void_p]),
    "hipad_accumulate_bf16": (c_int, [c_void_p, c_int, c_void_p]),
    "hipad_depth_loss_workspace": (c_size_t, []),
    "hipad_depth_loss_forward": (c_int, [c_void_p] * 4 + [c_size_t, c_void_p, ctypes.c_longlong, c_void_p, c_void_p]
                                 + [c_int] * 3 + [ctypes.c_float] * 3 + [c_void_p]),
    "hipad_depth_loss_backward": (c_int, [c_void_p] * 5 + [ctypes.c_longlong, c_void_p] + [c_int] * 3 + [ctypes.c_float, c_void_p]),
    "hipad_add_rows": (c_int, [c_void_p] * 5 + [c_int] * 3 + [c_void_p]),
    "hipad_rows_sum": (c_int, [c_void_p] * 2 + [c_int] * 3 + [c_void_p]),
    "hipad_step_offsets": (c_int, [c_void_p] * 2 + [ctypes.c_longlong] + [c_int] * 3 + [c_void_p]),
    "hipad_chain_forward": (c_int, [c_void_p, c_int, c_void_p]),
    "hipad_chain_debug_stamps": (None, [c_void_p]),
    "hipad_chain_backward_dx": (c_int, [c_void_p, c_int, c_void_p]),
    "hipad_chain_backward_dw": (c_int, [c_void_p, c_int, c_void_p]),
    "hipad_pack_weights": (c_int, [c_void_p] * 6 + [c_int, c_int, c_void_p]),
    "hipad_shadow_bf16": (c_int, [c_void_p, c_void_p, ctypes.c_longlong, c_void_p]),
    "hipad_attention_forward": (c_int, [c_void_p] * 5 + [c_int] * 5 + [ctypes.c_float, ctypes.c_float, ctypes.c_uint,
                                                                     c_void_p, c_void_p]),
    "hipad_attention_backward": (c_int, [c_void_p] * 10 + [c_int] * 5 + [ctypes.c_float, ctypes.c_float,
                                                                       ctypes.c_uint, c_void_p, c_void_p]),
}

_lib = None


class HipadError(RuntimeError):
    pass


class HipadLayoutError(HipadError):
    """An argument layout a kernel does not take, found on the host BEFORE anything was launched."""


def load():
    """dlopen libhipad.so and bind every declared symbol (raises if anything is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise HipadError(
            f"{SO_PATH} not found: build it with `python hip-ad_amd/build.py` "
            "(or __graft_entry__.build()); there is no fallback path")
    lib = ctypes.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().hipad_status_string(status).decode()
        raise HipadError(f"{what}: {msg} (status {status})")


def stream_ptr(device=None):
    return torch.cuda.current_stream(device).cuda_stream


def _req(t, dtype, name):
    if not t.is_cuda:
        raise HipadError(f"{name} must be a device tensor (got {t.device}); the hot path has no CPU fallback")
    if t.dtype != dtype:
        raise HipadError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise HipadError(f"{name} must be contiguous")
    return t


def daf_dims(feat, spatial_shape, loc, weights):
    if feat.dim() != 3 or spatial_shape.dim() != 3 or loc.dim() != 5 or weights.dim() != 6:
        raise HipadError("deformable_aggregation: bad ranks "
                         f"feat{tuple(feat.shape)} shape{tuple(spatial_shape.shape)} "
                         f"loc{tuple(loc.shape)} weights{tuple(weights.shape)}")
    bs, num_feat, C = feat.shape
    cams, L = spatial_shape.shape[:2]
    A, P = loc.shape[1:3]
    G = weights.shape[5]
    if tuple(loc.shape) != (bs, A, P, cams, 2) or tuple(weights.shape) != (bs, A, P, cams, L, G):
        raise HipadError(f"deformable_aggregation: inconsistent shapes loc{tuple(loc.shape)} "
                         f"weights{tuple(weights.shape)} for bs={bs} cams={cams} L={L}")
    return bs, cams, num_feat, C, L, A, P, G


_ws_cache = {}


def _workspace(nbytes, device):
    """Grow-only scratch per (device, stream): calls on one stream are serialised and may share it, calls issued
    on different streams (the decoder runs its modality branches concurrently) must not."""
    if torch.cuda.is_current_stream_capturing():
        # memory handed out during a capture belongs to that graph's private pool: caching it under the stream handle
        # would outlive the graph (torch recycles stream handles) and hand a later caller freed memory.  A fresh
        # buffer per call is the usual captured-temporary pattern: the pool reuses the block for the next call.
        return torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def daf_forward(feat, spatial_shape, scale_start_index, loc, weights, out=None):
    lib = load()
    bf16 = feat.dtype == torch.bfloat16        # the encoder's own rows: hipad_daf_forward_bf16 (same values, half the bytes)
    _req(feat, torch.bfloat16 if bf16 else torch.float32, "feat"); _req(loc, torch.float32, "sampling_location")
    _req(weights, torch.float32, "weights"); _req(spatial_shape, torch.int32, "spatial_shape")
    _req(scale_start_index, torch.int32, "scale_start_index")
    d = daf_dims(feat, spatial_shape, loc, weights)
    bs, cams, num_feat, C, L, A, P, G = d
    if out is None:
        out = torch.empty(bs, A, C, dtype=torch.float32, device=feat.device)
    nbytes = lib.hipad_daf_forward_workspace(*d)
    ws = _workspace(nbytes, feat.device) if nbytes else None
    with torch.cuda.device(feat.device):
        st = (lib.hipad_daf_forward_bf16 if bf16 else lib.hipad_daf_forward)(out.data_ptr(), feat.data_ptr(), spatial_shape.data_ptr(),
                                   scale_start_index.data_ptr(), loc.data_ptr(), weights.data_ptr(), *d,
                                   ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
                                   stream_ptr(feat.device))
    check(st, "hipad_daf_forward")
    return out


def daf_backward(feat, spatial_shape, scale_start_index, loc, weights, grad_out,
                 grad_feat=None, grad_loc=None, grad_w=None, overwrite_loc_w=False, atomic_feat=False):
    """Accumulates into grad_feat (always) and into / over grad_loc, grad_w (see include/hipad.h)."""
    lib = load()
    d = daf_dims(feat, spatial_shape, loc, weights)
    _req(grad_out, torch.float32, "grad_output")
    for t, n in ((grad_feat, "grad_feat"), (grad_loc, "grad_loc"), (grad_w, "grad_weights")):
        if t is not None:
            _req(t, torch.float32, n)
    flags = (1 if overwrite_loc_w else 0) | (2 if atomic_feat else 0)
    nbytes = 0 if (atomic_feat or grad_feat is None) else lib.hipad_daf_backward_workspace(*d)
    ws = _workspace(nbytes, feat.device) if nbytes else None
    bf16 = feat.dtype == torch.bfloat16
    if bf16 and atomic_feat:
        raise HipadError("daf_backward: bf16 feature rows have no atomic-scatter variant")
    with torch.cuda.device(feat.device):
        st = (lib.hipad_daf_backward_bf16 if bf16 else lib.hipad_daf_backward)(
            feat.data_ptr(), spatial_shape.data_ptr(), scale_start_index.data_ptr(), loc.data_ptr(),
            weights.data_ptr(), grad_out.data_ptr(),
            grad_feat.data_ptr() if grad_feat is not None else None,
            grad_loc.data_ptr() if grad_loc is not None else None,
            grad_w.data_ptr() if grad_w is not None else None,
            *d, flags, ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0,
            stream_ptr(feat.device))
    check(st, "hipad_daf_backward")


class DafCall(ctypes.Structure):
    """include/hipad.h: hipad_daf_call."""
    _fields_ = [("loc", c_void_p), ("weights", c_void_p), ("grad_out", c_void_p),
                ("num_anchors", ctypes.c_int32), ("num_pts", ctypes.c_int32)]


DAF_MAX_CALLS = 64


def daf_backward_feat_multi(calls, grad_feat, spatial_shape, scale_start_index):
    """grad_feat (bs, num_feat, 256) fp32 += the feature gradient of every (loc, weights, grad_out) triple of ``calls``
    -- hipad_daf_backward_feat_multi: one counting sort + one accumulation pass for all of them (tables of more than
    DAF_MAX_CALLS entries are worked off in several passes)."""
    lib = load()
    _req(grad_feat, torch.float32, "grad_feat"); _req(spatial_shape, torch.int32, "spatial_shape")
    _req(scale_start_index, torch.int32, "scale_start_index")
    bs, num_feat, C = grad_feat.shape
    cams, L = spatial_shape.shape[:2]
    dev = grad_feat.device
    for lo in range(0, len(calls), DAF_MAX_CALLS):
        part = calls[lo:lo + DAF_MAX_CALLS]
        table = (DafCall * len(part))()
        G = None
        for k, (loc, w, gout) in enumerate(part):
            _req(loc, torch.float32, "sampling_location"); _req(w, torch.float32, "weights"); _req(gout, torch.float32, "grad_output")
            A, P = loc.shape[1:3]
            G = w.shape[5]
            if tuple(loc.shape) != (bs, A, P, cams, 2) or tuple(w.shape) != (bs, A, P, cams, L, G) \
                    or tuple(gout.shape) != (bs, A, C):
                raise HipadError(f"daf_backward_feat_multi: call {lo + k} has loc{tuple(loc.shape)} weights{tuple(w.shape)} "
                                 f"grad_out{tuple(gout.shape)} for bs={bs} cams={cams} L={L} C={C}")
            table[k] = DafCall(loc.data_ptr(), w.data_ptr(), gout.data_ptr(), A, P)
        dims = (bs, cams, num_feat, C, L, G)
        nbytes = lib.hipad_daf_backward_feat_multi_workspace(ctypes.addressof(table), len(part), *dims)
        if not nbytes:
            raise HipadError(f"daf_backward_feat_multi: unsupported shapes {dims} ({len(part)} calls)")
        ws = _workspace(nbytes, dev)
        with torch.cuda.device(dev):
            st = lib.hipad_daf_backward_feat_multi(ctypes.addressof(table), len(part), grad_feat.data_ptr(),
                                                   spatial_shape.data_ptr(), scale_start_index.data_ptr(), *dims,
                                                   ws.data_ptr(), ws.numel(), stream_ptr(dev))
        check(st, "hipad_daf_backward_feat_multi")


def daf_taps(spatial_shape, scale_start_index, loc, num_feat):
    lib = load()
    bs, A, P, cams = loc.shape[:4]
    L = spatial_shape.shape[1]
    valid = torch.empty(bs, A, P, cams, dtype=torch.uint8, device=loc.device)
    taps = torch.empty(bs, A, P, cams, L, 4, dtype=torch.int32, device=loc.device)
    with torch.cuda.device(loc.device):
        st = lib.hipad_daf_taps(valid.data_ptr(), taps.data_ptr(), spatial_shape.data_ptr(),
                                scale_start_index.data_ptr(), loc.data_ptr(), bs, cams, num_feat, L, A, P,
                                stream_ptr(loc.device))
    check(st, "hipad_daf_taps")
    return valid, taps


def _ptr(t):
    return t.data_ptr() if t is not None else None


def project_points_forward(key_points, projection_mat, image_wh=None):
    """(bs,A,P,3) -> (bs,A,P,cams,2); see include/hipad.h."""
    lib = load()
    _req(key_points, torch.float32, "key_points"); _req(projection_mat, torch.float32, "projection_mat")
    if image_wh is not None:
        _req(image_wh, torch.float32, "image_wh")
    bs, A, P = key_points.shape[:3]
    cams = projection_mat.shape[1]
    if key_points.shape[-1] != 3 or tuple(projection_mat.shape) != (bs, cams, 4, 4):
        raise HipadError(f"project_points: bad shapes {tuple(key_points.shape)} {tuple(projection_mat.shape)}")
    loc = torch.empty(bs, A, P, cams, 2, dtype=torch.float32, device=key_points.device)
    with torch.cuda.device(key_points.device):
        st = lib.hipad_project_points_forward(loc.data_ptr(), key_points.data_ptr(), projection_mat.data_ptr(),
                                              _ptr(image_wh), bs, A, P, cams, stream_ptr(key_points.device))
    check(st, "hipad_project_points_forward")
    return loc


def project_points_backward(grad_loc, key_points, projection_mat, image_wh=None):
    lib = load()
    _req(grad_loc, torch.float32, "grad_loc")
    bs, A, P = key_points.shape[:3]
    cams = projection_mat.shape[1]
    gkp = torch.empty_like(key_points)
    with torch.cuda.device(key_points.device):
        st = lib.hipad_project_points_backward(gkp.data_ptr(), grad_loc.data_ptr(), key_points.data_ptr(),
                                               projection_mat.data_ptr(), _ptr(image_wh), bs, A, P, cams,
                                               stream_ptr(key_points.device))
    check(st, "hipad_project_points_backward")
    return gkp


def weights_softmax_forward(u, v, keep, L, P, G):
    """u (bs,A,n) [+ v (bs,cams,n)] or u (bs,A,cams,n) with v=None -> weights (bs,A,P,cams,L,G), stats."""
    lib = load()
    _req(u, torch.float32, "u")
    if v is not None:
        _req(v, torch.float32, "v")
    if keep is not None:
        _req(keep, torch.float32, "keep")
    n = L * P * G
    per_cam = u.dim() == 4
    bs, A = u.shape[:2]
    cams = u.shape[2] if per_cam else v.shape[1]
    if u.shape[-1] != n or (v is not None and tuple(v.shape) != (bs, cams, n)) or (not per_cam and v is None):
        raise HipadError(f"weights_softmax: bad shapes u{tuple(u.shape)} v{None if v is None else tuple(v.shape)} "
                         f"L={L} P={P} G={G}")
    w = torch.empty(bs, A, P, cams, L, G, dtype=torch.float32, device=u.device)
    stats = torch.empty(bs, A, G, 2, dtype=torch.float32, device=u.device)
    with torch.cuda.device(u.device):
        st = lib.hipad_weights_softmax_forward(w.data_ptr(), stats.data_ptr(), u.data_ptr(), _ptr(v),
                                               _ptr(keep), bs, A, cams, L, P, G, int(per_cam), stream_ptr(u.device))
    check(st, "hipad_weights_softmax_forward")
    return w, stats


def weights_softmax_backward(grad_w, stats, u, v, keep, L, P, G):
    lib = load()
    _req(grad_w, torch.float32, "grad_weights")
    per_cam = u.dim() == 4
    bs, A = u.shape[:2]
    cams = u.shape[2] if per_cam else v.shape[1]
    gu = torch.empty_like(u)
    gv = torch.empty_like(v) if v is not None else None
    nbytes = lib.hipad_weights_softmax_backward_workspace(bs, A, cams, L, P, G, int(v is not None))
    ws = _workspace(nbytes, u.device) if nbytes else None
    with torch.cuda.device(u.device):
        st = lib.hipad_weights_softmax_backward(gu.data_ptr(), _ptr(gv), grad_w.data_ptr(), stats.data_ptr(),
                                                u.data_ptr(), _ptr(v), _ptr(keep), bs, A, cams, L, P, G,
                                                int(per_cam), _ptr(ws), ws.numel() if ws is not None else 0,
                                                stream_ptr(u.device))
    check(st, "hipad_weights_softmax_backward")
    return gu, gv


def attention_forward(q, k, v, heads, scale, p_drop=0.0, seed=0, need_lse=True, seed_dev=None):
    """q (B,Nq,E), k/v (B,Nk,E), E = heads*D -> out (B,Nq,E), lse (B,heads,Nq) or None."""
    lib = load()
    _req(q, torch.float32, "q"); _req(k, torch.float32, "k"); _req(v, torch.float32, "v")
    B, Nq, E = q.shape
    Nk = k.shape[1]
    if E % heads or tuple(k.shape) != (B, Nk, E) or tuple(v.shape) != (B, Nk, E):
        raise HipadError(f"attention: bad shapes q{tuple(q.shape)} k{tuple(k.shape)} v{tuple(v.shape)} heads={heads}")
    out = torch.empty_like(q)
    lse = torch.empty(B, heads, Nq, dtype=torch.float32, device=q.device) if need_lse else None
    with torch.cuda.device(q.device):
        st = lib.hipad_attention_forward(out.data_ptr(), _ptr(lse), q.data_ptr(), k.data_ptr(), v.data_ptr(), B, heads,
                                         Nq, Nk, E // heads, float(scale), float(p_drop), int(seed) & 0xFFFFFFFF,
                                         _ptr(seed_dev), stream_ptr(q.device))
    check(st, "hipad_attention_forward")
    return out, lse


def attention_backward(dout, out, lse, q, k, v, heads, scale, p_drop=0.0, seed=0, seed_dev=None):
    lib = load()
    _req(dout, torch.float32, "dout")
    B, Nq, E = q.shape
    Nk = k.shape[1]
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(B, heads, Nq, dtype=torch.float32, device=q.device)
    with torch.cuda.device(q.device):
        st = lib.hipad_attention_backward(dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), delta.data_ptr(), dout.data_ptr(),
                                          out.data_ptr(), lse.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), B,
                                          heads, Nq, Nk, E // heads, float(scale), float(p_drop),
                                          int(seed) & 0xFFFFFFFF, _ptr(seed_dev), stream_ptr(q.device))
    check(st, "hipad_attention_backward")
    return dq, dk, dv


def linear_forward(x2, weight, bias, relu):
    """x2 (M,K) fp32 contiguous, weight (N,K) (rows contiguous), bias (N,) or None -> y (M,N)."""
    lib = load()
    M, K = x2.shape
    N = weight.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=x2.device)
    with torch.cuda.device(x2.device):
        st = lib.hipad_linear_forward(y.data_ptr(), x2.data_ptr(), weight.data_ptr(), _ptr(bias), M, N, K, int(relu),
                                      stream_ptr(x2.device))
    check(st, "hipad_linear_forward")
    return y


def linear_backward(dy2, y_relu, x2, weight, dx, dw, db):
    """dx overwritten (or None); dw / db accumulated into (or None)."""
    lib = load()
    M, K = x2.shape
    N = weight.shape[0]
    with torch.cuda.device(x2.device):
        st = lib.hipad_linear_backward(_ptr(dx), _ptr(dw), _ptr(db), dy2.data_ptr(), _ptr(y_relu), x2.data_ptr(),
                                       weight.data_ptr(), M, N, K, stream_ptr(x2.device))
    check(st, "hipad_linear_backward")


class LrScheduleStruct(ctypes.Structure):
    """hipad_lr_schedule of include/hipad.h."""
    _fields_ = [("policy", c_int), ("warmup_iters", c_int), ("warmup_ratio", ctypes.c_float), ("max_iters", c_int),
                ("min_lr_ratio", ctypes.c_float)]


def lr_factor(sched, iteration):
    """Host evaluation of the kernel's learning-rate factor (hipad_lr_factor); ``sched``: LrScheduleStruct or None."""
    return float(load().hipad_lr_factor(None if sched is None else ctypes.byref(sched), int(iteration)))


def shadow_bf16(dst, src):
    """dst (bf16, n) <- src (fp32, n), element order unchanged."""
    lib = load()
    _req(src, torch.float32, "src")
    _req(dst, torch.bfloat16, "dst")
    if dst.numel() != src.numel():
        raise HipadError("shadow_bf16: lengths differ")
    with torch.cuda.device(src.device):
        check(lib.hipad_shadow_bf16(dst.data_ptr(), src.data_ptr(), src.numel(), stream_ptr(src.device)), "hipad_shadow_bf16")


def adamw_step(param, grad, exp_avg, exp_avg_sq, n_group0, lr0, lr1, betas, eps, weight_decay, max_norm, step_dev,
               norm_out, workspace, zero_grad=True, sched=None, shadow=None):
    """Clip (global norm) + AdamW over flat fp32 buffers; see include/hipad.h."""
    lib = load()
    for t, nm in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _req(t, torch.float32, nm)
    n = param.numel()
    if not (grad.numel() == exp_avg.numel() == exp_avg_sq.numel() == n):
        raise HipadError("adamw_step: buffers differ in length")
    if step_dev.dtype != torch.int32 or not step_dev.is_cuda:
        raise HipadError("adamw_step: step_dev must be a device int32 tensor")
    if norm_out is not None and (norm_out.numel() < 2 or norm_out.dtype != torch.float32):
        raise HipadError("adamw_step: norm_out must hold two floats (norm, learning rate)")
    if shadow is not None:
        _req(shadow, torch.bfloat16, "shadow")
        if shadow.numel() != n:
            raise HipadError("adamw_step: shadow buffer differs in length")
    with torch.cuda.device(param.device):
        st = lib.hipad_adamw_step(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), n,
                                  int(n_group0), float(lr0), float(lr1), float(betas[0]), float(betas[1]), float(eps),
                                  float(weight_decay), float(max_norm if max_norm else 0.0), step_dev.data_ptr(),
                                  _ptr(norm_out), workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                  int(bool(zero_grad)), None if sched is None else ctypes.addressof(sched),
                                  _ptr(shadow), stream_ptr(param.device))
    check(st, "hipad_adamw_step")


def layernorm_forward(x2, gamma, beta, eps, need_stats=True):
    """x2 (M,N) fp32 contiguous -> y, mean, rstd."""
    lib = load()
    _req(x2, torch.float32, "x")
    M, N = x2.shape
    y = torch.empty_like(x2)
    mean = torch.empty(M, dtype=torch.float32, device=x2.device) if need_stats else None
    rstd = torch.empty(M, dtype=torch.float32, device=x2.device) if need_stats else None
    with torch.cuda.device(x2.device):
        st = lib.hipad_layernorm_forward(y.data_ptr(), _ptr(mean), _ptr(rstd), x2.data_ptr(), _ptr(gamma), _ptr(beta), M, N,
                                         float(eps), stream_ptr(x2.device))
    check(st, "hipad_layernorm_forward")
    return y, mean, rstd


def layernorm_backward(dy2, x2, mean, rstd, gamma, dx, dgamma, dbeta):
    """dx overwritten (or None); dgamma / dbeta accumulated into (or None)."""
    lib = load()
    _req(dy2, torch.float32, "dy")
    M, N = x2.shape
    with torch.cuda.device(x2.device):
        st = lib.hipad_layernorm_backward(_ptr(dx), _ptr(dgamma), _ptr(dbeta), dy2.data_ptr(), x2.data_ptr(), mean.data_ptr(),
                                          rstd.data_ptr(), _ptr(gamma), M, N, stream_ptr(x2.device))
    check(st, "hipad_layernorm_backward")


def linear_assignment(cost, n_rows):
    """cost (B, R, C) fp32 (rows = ground truth, cols = predictions), n_rows (B,) int32 -> col_of_row (B, R) int32."""
    lib = load()
    _req(cost, torch.float32, "cost")
    _req(n_rows, torch.int32, "n_rows")
    B, R, C = cost.shape
    if n_rows.numel() != B:
        raise HipadError("linear_assignment: n_rows must have one entry per problem")
    out = torch.empty(B, R, dtype=torch.int32, device=cost.device)
    with torch.cuda.device(cost.device):
        st = lib.hipad_linear_assignment(out.data_ptr(), cost.data_ptr(), n_rows.data_ptr(), B, R, C, stream_ptr(cost.device))
    check(st, "hipad_linear_assignment")
    return out


def linear_relu_ln_supported(x2, weight):
    lib = load()
    return bool(lib.hipad_linear_relu_ln_supported(weight.shape[0], weight.shape[1])) and x2.data_ptr() % 16 == 0 \
        and weight.data_ptr() % 16 == 0


def linear_relu_ln_forward(x2, weight, bias, gamma, beta, eps):
    """x2 (M,K) -> y = LayerNorm(relu(x2 W^T + b)) (M,N), x_relu (M,N), mean (M,), rstd (M,)."""
    lib = load()
    _req(x2, torch.float32, "x")
    M, K = x2.shape
    N = weight.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=x2.device)
    xr = torch.empty_like(y)
    mean = torch.empty(M, dtype=torch.float32, device=x2.device)
    rstd = torch.empty_like(mean)
    with torch.cuda.device(x2.device):
        st = lib.hipad_linear_relu_ln_forward(y.data_ptr(), xr.data_ptr(), mean.data_ptr(), rstd.data_ptr(), x2.data_ptr(),
                                              weight.data_ptr(), _ptr(bias), _ptr(gamma), _ptr(beta), M, N, K, float(eps),
                                              stream_ptr(x2.device))
    check(st, "hipad_linear_relu_ln_forward")
    return y, xr, mean, rstd


def box_points_project_forward(anchor, fix_scale, learn, projection_mat, image_wh, want_key_points=False):
    """anchor (bs,A,D) -> loc (bs,A,P,cams,2) [+ key_points (bs,A,P,3)]; see include/hipad.h."""
    lib = load()
    _req(anchor, torch.float32, "anchor"); _req(fix_scale, torch.float32, "fix_scale")
    _req(projection_mat, torch.float32, "projection_mat")
    bs, A, D = anchor.shape
    n_fix = fix_scale.shape[0]
    n_learn = 0 if learn is None else learn.shape[-1] // 3
    cams = projection_mat.shape[1]
    P = n_fix + n_learn
    loc = torch.empty(bs, A, P, cams, 2, dtype=torch.float32, device=anchor.device)
    kp = torch.empty(bs, A, P, 3, dtype=torch.float32, device=anchor.device) if want_key_points else None
    with torch.cuda.device(anchor.device):
        st = lib.hipad_box_points_project_forward(loc.data_ptr(), _ptr(kp), anchor.data_ptr(), fix_scale.data_ptr(), _ptr(learn),
                                                  projection_mat.data_ptr(), _ptr(image_wh), bs, A, n_fix, n_learn, cams, D,
                                                  stream_ptr(anchor.device))
    check(st, "hipad_box_points_project_forward")
    return loc, kp


def box_points_project_backward(grad_loc, anchor, fix_scale, learn, projection_mat, image_wh):
    lib = load()
    _req(grad_loc, torch.float32, "grad_loc")
    bs, A, D = anchor.shape
    n_fix = fix_scale.shape[0]
    n_learn = 0 if learn is None else learn.shape[-1] // 3
    cams = projection_mat.shape[1]
    g_anchor = torch.empty_like(anchor)
    g_learn = torch.empty_like(learn) if learn is not None else None
    with torch.cuda.device(anchor.device):
        st = lib.hipad_box_points_project_backward(g_anchor.data_ptr(), _ptr(g_learn), grad_loc.data_ptr(), anchor.data_ptr(),
                                                   fix_scale.data_ptr(), _ptr(learn), projection_mat.data_ptr(), _ptr(image_wh),
                                                   bs, A, n_fix, n_learn, cams, D, stream_ptr(anchor.device))
    check(st, "hipad_box_points_project_backward")
    return g_anchor, g_learn


def line_points_project_forward(anchor, offset, heights, projection_mat, image_wh, S, Hn, K):
    lib = load()
    for t, n in ((anchor, "anchor"), (offset, "offset"), (heights, "heights"), (projection_mat, "projection_mat")):
        _req(t, torch.float32, n)
    bs, A = anchor.shape[:2]
    cams = projection_mat.shape[1]
    loc = torch.empty(bs, A, S * Hn * K, cams, 2, dtype=torch.float32, device=anchor.device)
    with torch.cuda.device(anchor.device):
        st = lib.hipad_line_points_project_forward(loc.data_ptr(), anchor.data_ptr(), offset.data_ptr(), heights.data_ptr(),
                                                   projection_mat.data_ptr(), _ptr(image_wh), bs, A, S, Hn, K, cams,
                                                   stream_ptr(anchor.device))
    check(st, "hipad_line_points_project_forward")
    return loc


def line_points_project_backward(grad_loc, anchor, offset, heights, projection_mat, image_wh, S, Hn, K):
    lib = load()
    _req(grad_loc, torch.float32, "grad_loc")
    bs, A = anchor.shape[:2]
    cams = projection_mat.shape[1]
    g_anchor, g_offset = torch.empty_like(anchor), torch.empty_like(offset)
    with torch.cuda.device(anchor.device):
        st = lib.hipad_line_points_project_backward(g_anchor.data_ptr(), g_offset.data_ptr(), grad_loc.data_ptr(),
                                                    anchor.data_ptr(), offset.data_ptr(), heights.data_ptr(),
                                                    projection_mat.data_ptr(), _ptr(image_wh), bs, A, S, Hn, K, cams,
                                                    stream_ptr(anchor.device))
    check(st, "hipad_line_points_project_backward")
    return g_anchor, g_offset


def focal_loss_forward(logits, target, weight, avg_factor, layers, alpha, gamma):
    """logits (rows, C) fp32, target (rows,) int64 -> loss_per_layer (layers,), grad_logits (rows, C)."""
    lib = load()
    _req(logits, torch.float32, "logits"); _req(target, torch.int64, "target")
    rows, C = logits.shape
    loss = torch.empty(layers, dtype=torch.float32, device=logits.device)
    grad = torch.empty_like(logits)
    with torch.cuda.device(logits.device):
        st = lib.hipad_focal_loss_forward(loss.data_ptr(), grad.data_ptr(), logits.data_ptr(), target.data_ptr(), _ptr(weight),
                                          _ptr(avg_factor), rows, C, int(layers), float(alpha), float(gamma),
                                          stream_ptr(logits.device))
    check(st, "hipad_focal_loss_forward")
    return loss, grad


BN_REPLICAS = 4
BN_SUM_FLOATS = 2       # a partial sum is one 64-bit fixed-point word = two float slots of the scratch buffer


def bn_supported(rows, channels):
    return bool(load().hipad_bn_supported(int(rows), int(channels)))


def bn_forward(x, residual, gamma, beta, running_mean, running_var, sums, eps, momentum, relu, out=None):
    """x (N, C, H, W) bf16 channels-last -> (y like x, save (2C,) fp32).  ``sums``: zeroed scratch of BN_REPLICAS * 2C * BN_SUM_FLOATS
    floats (64-bit fixed-point partial sums, see include/hipad.h).  ``out``: (groups, rows_per_group, C) bf16 view whose
    groups may be strided (a level's block inside the flat pyramid, one group per sample): y is written there instead
    of into a fresh tensor and ``out`` is returned as y."""
    lib = load()
    n, c, h, w = x.shape
    rows = n * h * w
    save = torch.empty(2 * c, dtype=torch.float32, device=x.device)
    if out is None:
        y = torch.empty_like(x)           # preserves channels-last
        group_rows = group_stride = rows
    else:
        if out.dtype != x.dtype or out.dim() != 3 or out.shape[2] != c or out.shape[0] * out.shape[1] != rows \
                or out.stride(2) != 1 or out.stride(1) != c or out.stride(0) % c or out.stride(0) < out.shape[1] * c:
            raise HipadError(f"bn_forward: out {tuple(out.shape)} / strides {out.stride()} does not hold {rows} rows of {c}")
        y, group_rows, group_stride = out, out.shape[1], out.stride(0) // c
    with torch.cuda.device(x.device):
        check(lib.hipad_bn_forward_grouped(y.data_ptr(), save.data_ptr(), sums.data_ptr(), x.data_ptr(), _ptr(residual),
                                           gamma.data_ptr(), beta.data_ptr(), _ptr(running_mean), _ptr(running_var), rows, c,
                                           float(eps), float(momentum), int(bool(relu)), group_rows, group_stride,
                                           stream_ptr(x.device)), "hipad_bn_forward")
    return y, save


def bn_backward(dy, y, x, save, gamma, gsums, dgamma, dbeta, need_res):
    """-> (dx, dres or None), bf16 channels-last like x; dgamma / dbeta (fp32, may be None) are accumulated into."""
    lib = load()
    n, c, h, w = x.shape
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if need_res else None
    with torch.cuda.device(x.device):
        check(lib.hipad_bn_backward(dx.data_ptr(), _ptr(dres), _ptr(dgamma), _ptr(dbeta), gsums.data_ptr(), dy.data_ptr(), _ptr(y),
                                    x.data_ptr(), save.data_ptr(), gamma.data_ptr(), n * h * w, c, stream_ptr(x.device)),
              "hipad_bn_backward")
    return dx, dres


def grid_mask(x, params, use_h, use_w, mode, out_dtype=torch.float32, channels_last=False):
    """x (n, c, h, w) fp32 * stripe mask of ``params`` = [apply, d, l, st_h, st_w] (device) -> ``out_dtype`` tensor."""
    lib = load()
    _req(x, torch.float32, "x"); _req(params, torch.float32, "params")
    if out_dtype not in (torch.float32, torch.bfloat16):
        raise HipadError("grid_mask: out_dtype must be float32 or bfloat16")
    n, c, h, w = x.shape
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device,
                      memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    strides = (ctypes.c_longlong * 4)(*out.stride())
    with torch.cuda.device(x.device):
        check(lib.hipad_grid_mask(out.data_ptr(), int(out_dtype == torch.bfloat16), strides, x.data_ptr(), params.data_ptr(),
                                  n, c, h, w, int(bool(use_h)), int(bool(use_w)), int(mode), stream_ptr(x.device)),
              "hipad_grid_mask")
    return out


def dropout_add(x, base, p_drop, seed, seed_dev):
    """base + dropout(x) (base None: dropout(x) alone, i.e. the backward on an output gradient); fp32 contiguous."""
    lib = load()
    _req(x, torch.float32, "x")
    if base is not None:
        _req(base, torch.float32, "base")
        if base.shape != x.shape:
            raise HipadError("dropout_add: shapes differ")
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        check(lib.hipad_dropout_add(out.data_ptr(), x.data_ptr(), _ptr(base), x.numel(), float(p_drop), int(seed) & 0xFFFFFFFF,
                                    _ptr(seed_dev), stream_ptr(x.device)), "hipad_dropout_add")
    return out


def chunk_mix(x0, x1, table, rows):
    """out chunk g = sum_k table[g][k] * (x0 chunk k + x1 chunk k); x: (bs, K * rows, C) -> (bs, G * rows, C).
    ``table``: G x K nested list / tuple of floats (host)."""
    lib = load()
    _req(x0, torch.float32, "x0")
    if x1 is not None:
        _req(x1, torch.float32, "x1")
        if x1.shape != x0.shape:
            raise HipadError("chunk_mix: x0 / x1 shapes differ")
    G, K = len(table), len(table[0])
    bs, n, C = x0.shape
    if n != K * rows:
        raise HipadError(f"chunk_mix: {n} rows != {K} chunks x {rows}")
    out = torch.empty(bs, G * rows, C, dtype=torch.float32, device=x0.device)
    flat = (ctypes.c_float * (G * K))(*[float(v) for row in table for v in row])
    with torch.cuda.device(x0.device):
        check(lib.hipad_chunk_mix(out.data_ptr(), x0.data_ptr(), _ptr(x1), flat, bs, K, G, rows, C, stream_ptr(x0.device)),
              "hipad_chunk_mix")
    return out


def motion_query_embed(cls, box, table, freq, sin_col, cos_col):
    """(bs, A, ncls) logits, (bs, A, D) boxes, (ncls, modes, ts, 2) anchors, (half,) frequencies -> (bs, A, modes, 2*half)."""
    lib = load()
    for t, n in ((cls, "cls"), (box, "box"), (table, "table"), (freq, "freq")):
        _req(t, torch.float32, n)
    bs, A, ncls = cls.shape
    D = box.shape[-1]
    modes, ts = table.shape[1], table.shape[2]
    half = freq.numel()
    out = torch.empty(bs, A, modes, 2 * half, dtype=torch.float32, device=cls.device)
    with torch.cuda.device(cls.device):
        check(lib.hipad_motion_query_embed(out.data_ptr(), cls.data_ptr(), box.data_ptr(), table.data_ptr(), freq.data_ptr(),
                                           bs * A, ncls, D, sin_col, cos_col, modes, ts, half, stream_ptr(cls.device)),
              "hipad_motion_query_embed")
    return out


class DepthLevel(ctypes.Structure):
    _fields_ = [("gt", c_void_p), ("weight", c_void_p), ("bias", c_void_p), ("grad_weight", c_void_p), ("grad_bias", c_void_p),
                ("rows_per_cam", ctypes.c_int32), ("row_offset", ctypes.c_int32)]


DEPTH_MAX_LEVELS = 4


def _depth_levels(feat, levels, cams, grads=None):
    """levels: [(gt fp32, weight fp32 (256 values), bias fp32 (1 value), rows_per_cam, row_offset)] -> ctypes array."""
    if feat.dim() != 3 or feat.shape[-1] != 256 or feat.dtype != torch.bfloat16 or not feat.is_cuda or not feat.is_contiguous():
        raise HipadLayoutError("depth_loss: feat must be a contiguous bf16 CUDA tensor (bs, rows, 256)")
    if not 1 <= len(levels) <= DEPTH_MAX_LEVELS:
        raise HipadLayoutError(f"depth_loss: 1..{DEPTH_MAX_LEVELS} levels")
    bs = feat.shape[0]
    arr = (DepthLevel * len(levels))()
    for i, (gt, w, b, rpc, off) in enumerate(levels):
        for t, n, numel in ((gt, "gt", bs * cams * rpc), (w, "weight", 256), (b, "bias", 1)):
            _req(t, torch.float32, f"level {i} {n}")
            if t.numel() != numel:
                raise HipadLayoutError(f"depth_loss: level {i} {n} has {t.numel()} elements, expected {numel}")
        if off < 0 or off + cams * rpc > feat.shape[1]:
            raise HipadLayoutError(f"depth_loss: level {i} rows [{off}, {off + cams * rpc}) outside the pyramid")
        a = arr[i]
        a.gt, a.weight, a.bias = gt.data_ptr(), w.data_ptr(), b.data_ptr()
        a.rows_per_cam, a.row_offset = int(rpc), int(off)
        if grads is not None:
            gw, gb = grads[i]
            for t, n, numel in ((gw, "grad_weight", 256), (gb, "grad_bias", 1)):
                if t is not None:
                    _req(t, torch.float32, f"level {i} {n}")
                    if t.numel() != numel:
                        raise HipadLayoutError(f"depth_loss: level {i} {n} has {t.numel()} elements, expected {numel}")
            a.grad_weight, a.grad_bias = _ptr(gw), _ptr(gb)
    return arr


def depth_loss_forward(feat, focal, levels, cams, equal_focal, max_depth, loss_weight):
    """-> (loss (1 + L,): total then per level, coef (4,), pred (rows of all levels,)); see include/hipad.h."""
    lib = load()
    arr = _depth_levels(feat, levels, cams)
    bs = feat.shape[0]
    if focal is not None:
        _req(focal, torch.float32, "focal")
        if focal.numel() != bs * cams:
            raise HipadLayoutError("depth_loss: focal must hold bs * cams values")
    dev = feat.device
    total = sum(bs * cams * lv[3] for lv in levels)
    loss = torch.empty(1 + len(levels), dtype=torch.float32, device=dev)
    coef = torch.empty(DEPTH_MAX_LEVELS, dtype=torch.float32, device=dev)
    pred = torch.empty(total, dtype=torch.float32, device=dev)
    ws = torch.empty(lib.hipad_depth_loss_workspace() // 8, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        check(lib.hipad_depth_loss_forward(loss.data_ptr(), coef.data_ptr(), pred.data_ptr(), ws.data_ptr(), ws.numel() * 8,
                                           feat.data_ptr(), feat.shape[1], _ptr(focal), arr, len(levels), bs, cams,
                                           float(equal_focal), float(max_depth), float(loss_weight), stream_ptr(dev)),
              "hipad_depth_loss_forward")
    return loss, coef, pred


def depth_loss_backward(grad_feat, pred, coef, upstream, feat, levels, grads, cams, max_depth):
    """grad_feat (bs, rows, 256) fp32 += the loss's feature gradient; grads: [(grad_weight or None, grad_bias or None)]."""
    lib = load()
    arr = _depth_levels(feat, levels, cams, grads)
    _req(grad_feat, torch.float32, "grad_feat")
    if grad_feat.shape != feat.shape:
        raise HipadLayoutError("depth_loss: grad_feat must have the pyramid's shape")
    for t, n in ((pred, "pred"), (coef, "coef")):
        _req(t, torch.float32, n)
    if upstream is not None:
        _req(upstream, torch.float32, "upstream")
    with torch.cuda.device(feat.device):
        check(lib.hipad_depth_loss_backward(grad_feat.data_ptr(), pred.data_ptr(), coef.data_ptr(), _ptr(upstream), feat.data_ptr(),
                                            feat.shape[1], arr, len(levels), feat.shape[0], cams, float(max_depth),
                                            stream_ptr(feat.device)), "hipad_depth_loss_backward")


class AccItem(ctypes.Structure):
    _fields_ = [("dst", c_void_p), ("src", c_void_p), ("sizes", ctypes.c_int32 * 4), ("strides", ctypes.c_int32 * 4)]


ACC_MAX = 64


def accumulate_bf16(pairs):
    """``pairs``: [(dst fp32 contiguous, src bf16 of the same shape, any strides, <= 4-D)]: dst += src, 64 tensors a launch."""
    lib = load()
    if not pairs:
        return
    dev = pairs[0][0].device
    items = []
    for dst, src in pairs:
        try:
            _req(dst, torch.float32, "dst")
        except HipadError as e:
            raise HipadLayoutError(str(e)) from None
        if not src.is_cuda or src.dtype != torch.bfloat16 or src.device != dev or dst.device != dev:
            raise HipadLayoutError("accumulate_bf16: src must be a bf16 tensor on dst's device")
        if src.shape != dst.shape or src.dim() > 4 or src.numel() == 0:
            raise HipadLayoutError(f"accumulate_bf16: shapes {tuple(dst.shape)} / {tuple(src.shape)} (need equal, <= 4-D, non-empty)")
        # the kernel trusts the strides: the furthest element they reach must lie inside the source's storage
        last = src.storage_offset() + sum((n - 1) * st for n, st in zip(src.shape, src.stride()))
        if any(st < 0 for st in src.stride()) or last * 2 >= src.untyped_storage().nbytes():
            raise HipadLayoutError("accumulate_bf16: source strides reach outside its storage")
        pad = 4 - src.dim()
        it = AccItem()
        it.dst, it.src = dst.data_ptr(), src.data_ptr()
        for d in range(4):
            it.sizes[d] = 1 if d < pad else src.shape[d - pad]
            it.strides[d] = 0 if d < pad else src.stride(d - pad)
        items.append(it)
    with torch.cuda.device(dev):
        for i in range(0, len(items), ACC_MAX):
            chunk = items[i:i + ACC_MAX]
            arr = (AccItem * len(chunk))(*chunk)
            check(lib.hipad_accumulate_bf16(arr, len(chunk), stream_ptr(dev)), "hipad_accumulate_bf16")


def add_rows(base, rows):
    """base (bs, N, C) + every (bs, C) tensor of ``rows`` (1..3) broadcast over the N rows, one launch."""
    lib = load()
    _req(base, torch.float32, "base")
    if base.dim() != 3 or not 1 <= len(rows) <= 3:
        raise HipadError("add_rows: base must be (bs, N, C) and 1..3 row tensors given")
    bs, N, C = base.shape
    for k, r in enumerate(rows):
        _req(r, torch.float32, f"rows[{k}]")
        if r.numel() != bs * C:
            raise HipadError(f"add_rows: rows[{k}] has {r.numel()} elements, expected {bs} x {C}")
    out = torch.empty_like(base)
    ptrs = [r.data_ptr() for r in rows] + [None] * (3 - len(rows))
    with torch.cuda.device(base.device):
        check(lib.hipad_add_rows(out.data_ptr(), base.data_ptr(), ptrs[0], ptrs[1], ptrs[2], bs, N, C,
                                 stream_ptr(base.device)), "hipad_add_rows")
    return out


def rows_sum(x):
    """(bs, N, C) -> (bs, C): sum over the rows of every sample, fixed order."""
    lib = load()
    _req(x, torch.float32, "x")
    if x.dim() != 3 or x.numel() == 0:
        raise HipadError("rows_sum: x must be (bs, N, C), non-empty")
    bs, N, C = x.shape
    out = torch.empty(bs, C, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        check(lib.hipad_rows_sum(out.data_ptr(), x.data_ptr(), bs, N, C, stream_ptr(x.device)), "hipad_rows_sum")
    return out


def step_offsets(x, adjoint=False):
    """(..., steps, dims) way-points -> per-step offsets along the second-to-last axis (adjoint: the transposed map)."""
    lib = load()
    _req(x, torch.float32, "x")
    if x.dim() < 2 or x.numel() == 0:
        raise HipadError("step_offsets: x must be (..., steps, dims), non-empty")
    steps, dims = x.shape[-2], x.shape[-1]
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        check(lib.hipad_step_offsets(out.data_ptr(), x.data_ptr(), x.numel() // (steps * dims), steps, dims,
                                     1 if adjoint else 0, stream_ptr(x.device)), "hipad_step_offsets")
    return out


def keep_mask(shape, p_drop, seed, seed_dev, device):
    """Bernoulli keep mask scaled by 1 / (1 - p) in one launch; ``seed_dev``: device int32 step counter or None."""
    lib = load()
    out = torch.empty(shape, dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        check(lib.hipad_keep_mask(out.data_ptr(), out.numel(), float(p_drop), int(seed) & 0xFFFFFFFF, _ptr(seed_dev),
                                  stream_ptr(device)), "hipad_keep_mask")
    return out
