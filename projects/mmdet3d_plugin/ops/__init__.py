"""``projects.mmdet3d_plugin.ops`` -- the operator surface of the hot path.

Public symbols (same names, arguments and results as the reference's ops/__init__.py):
  deformable_aggregation_function(feature_maps, spatial_shape, scale_start_index,
                                  sampling_location, weights) -> (bs, num_anchors, C)
      reference: ops/__init__.py:7-30
  feature_maps_format(feature_maps, inverse=False)
      reference: ops/__init__.py:33-103
plus ``shared_feature_grad`` (ours) -- see deformable_aggregation.py.

One code path: the reference's per-device dispatch (an A800-specific build of the same
sources, ops/__init__.py:14-22) collapses to the gfx950 library.
"""
import torch

from .deformable_aggregation import DeformableAggregationFunction, shared_feature_grad

__all__ = ["deformable_aggregation_function", "feature_maps_format", "shared_feature_grad",
           "DeformableAggregationFunction", "level_major_tables"]


def deformable_aggregation_function(feature_maps, spatial_shape, scale_start_index, sampling_location, weights):
    token = getattr(feature_maps, "_hipad_grad_token", None)
    if token is not None and torch.is_grad_enabled():
        return DeformableAggregationFunction.apply(
            feature_maps, spatial_shape, scale_start_index, sampling_location, weights, token)
    return DeformableAggregationFunction.apply(
        feature_maps, spatial_shape, scale_start_index, sampling_location, weights)


# ------------------------------------------------------------------------------------------
# feature pyramid <-> flat "column" layout
#   col_feats [bs, sum_cam sum_level h*w, C]: camera-major, then level, then row-major (h, w),
#   channels last; spatial_shape [cams, L, 2] int64 (h, w); scale_start_index [cams, L] int64 =
#   exclusive prefix sum of h*w over the flattened (cam, level) axis.
# ------------------------------------------------------------------------------------------
def _tables(level_hw, num_cams, device):
    key = (tuple(level_hw), num_cams, str(device))
    hit = _tables.cache.get(key)
    if hit is None:
        ss = torch.tensor([list(map(list, level_hw))] * num_cams, dtype=torch.int64)
        sizes = (ss[..., 0] * ss[..., 1]).reshape(-1)
        start = (sizes.cumsum(0) - sizes).reshape(num_cams, len(level_hw))
        ss_d, start_d = ss.to(device), start.to(device)
        # host mirrors + int32 twins ride along so later calls need no device->host sync / cast
        ss_d._hipad_host = ss.tolist()
        ss_d._hipad_i32 = ((ss_d.data_ptr(), ss_d._version), ss_d.int())        # see deformable_aggregation._as_i32
        start_d._hipad_i32 = ((start_d.data_ptr(), start_d._version), start_d.int())
        hit = (ss_d, start_d)
        _tables.cache[key] = hit
    return hit


_tables.cache = {}


def level_major_tables(level_hw, num_cams, device):
    """(spatial_shape, scale_start_index, rows per sample, [(first row, rows)] per level) of a flat pyramid laid out LEVEL
    by level -- all cameras of level 0, then of level 1, ... -- which is how the encoder's level tensors sit in memory
    when its last norm layers write them in place (SparseDetector.extract_feat).  The aggregation operator reads
    positions through (spatial_shape, scale_start_index) only, so any consistent layout serves it; the reference's own
    ``feature_maps_format`` layout is camera-major (ops/__init__.py:78-96) and stays what ``feature_maps_format`` builds."""
    key = ("level-major", tuple(level_hw), num_cams, str(device))
    hit = _tables.cache.get(key)
    if hit is None:
        ss = torch.tensor([list(map(list, level_hw))] * num_cams, dtype=torch.int64)
        blocks, start, off = [], torch.zeros(num_cams, len(level_hw), dtype=torch.int64), 0
        for l, (h, w) in enumerate(level_hw):
            blocks.append((off, num_cams * h * w))
            start[:, l] = off + torch.arange(num_cams) * (h * w)
            off += num_cams * h * w
        ss_d, start_d = ss.to(device), start.to(device)
        ss_d._hipad_host = ss.tolist()
        ss_d._hipad_i32 = ((ss_d.data_ptr(), ss_d._version), ss_d.int())
        start_d._hipad_i32 = ((start_d.data_ptr(), start_d._version), start_d.int())
        hit = _tables.cache[key] = (ss_d, start_d, off, blocks)
    return hit


def _format_one_group(level_maps, out_dtype=None):
    bs, num_cams, C = level_maps[0].shape[:3]
    level_hw = [tuple(int(v) for v in m.shape[-2:]) for m in level_maps]
    per_cam = sum(h * w for h, w in level_hw)
    first = level_maps[0]
    # out_dtype: the encoder hands over bf16 channels-last levels; the copy below then converts to the fp32 the
    # aggregation kernels read in the same pass (no separate .float() of every level)
    col = torch.empty(bs, num_cams, per_cam, C, dtype=out_dtype or first.dtype, device=first.device)
    off = 0
    for m, (h, w) in zip(level_maps, level_hw):
        # (bs,cams,C,h,w) -> (bs,cams,h*w,C) written straight into its slot (no cat pass);
        # for channels-last producers this is a plain strided copy
        col[:, :, off:off + h * w].copy_(m.reshape(bs, num_cams, C, h * w).transpose(2, 3))
        off += h * w
    ss, start = _tables(level_hw, num_cams, first.device)
    return col.view(bs, num_cams * per_cam, C), ss, start


def _merge_groups(parts):
    col = torch.cat([p[0] for p in parts], dim=1)
    ss = torch.cat([p[1] for p in parts], dim=0)
    sizes = (ss[..., 0] * ss[..., 1]).reshape(-1)
    start = (sizes.cumsum(0) - sizes).reshape(ss.shape[0], ss.shape[1])
    host = []
    for p in parts:
        host += p[1]._hipad_host
    ss._hipad_host = host
    return col, ss, start


def _inverse(col_feats, spatial_shape, scale_start_index):
    levels = getattr(col_feats, "_hipad_levels", None)
    if levels is not None:
        # the detector's own level tensors ride on the flat tensor it made from them (same values): handing them back
        # keeps the consumers' gradient on the (small) levels instead of a slice of a pyramid-sized zero tensor
        return [list(levels)]
    host = getattr(spatial_shape, "_hipad_host", None)
    if host is None:
        host = spatial_shape.tolist()  # foreign tensor: one device->host copy
    bs, _, C = col_feats.shape
    groups = []  # runs of consecutive cameras with identical level shapes
    for cam, shapes in enumerate(host):
        if groups and groups[-1][0] == shapes:
            groups[-1][1] += 1
        else:
            groups.append([shapes, 1])
    out, row = [], 0
    for shapes, ncam in groups:
        per_cam = sum(h * w for h, w in shapes)
        block = col_feats[:, row:row + ncam * per_cam].reshape(bs, ncam, per_cam, C)
        maps, off = [], 0
        for h, w in shapes:
            maps.append(block[:, :, off:off + h * w].reshape(bs, ncam, h, w, C).permute(0, 1, 4, 2, 3))
            off += h * w
        out.append(maps)
        row += ncam * per_cam
    return out


def feature_maps_format(feature_maps, inverse=False, out_dtype=None):
    """``out_dtype`` (ours, optional): dtype of the flat tensor when it should differ from the levels' (the
    reference casts the levels to fp32 first, sparse_detector.py:84-89)."""
    if inverse:
        return _inverse(*feature_maps)
    if isinstance(feature_maps[0], (list, tuple)):
        return list(_merge_groups([_format_one_group(g, out_dtype) for g in feature_maps]))
    return list(_format_one_group(feature_maps, out_dtype))
