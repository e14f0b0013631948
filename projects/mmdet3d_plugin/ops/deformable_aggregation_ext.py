"""Stand-in for the reference's compiled extension module ``deformable_aggregation_ext``
(reference: ops/src/deformable_aggregation.cpp:127-138, a pybind11 module built from CUDA).

Same module name and the same two callables with the same argument order and semantics;
the work is done by hand-written gfx950 kernels in libhipad.so reached through its C ABI
(include/hipad.h), launched on torch's current stream.  There is no CPU fallback: CPU
tensors raise.
"""
import torch

from hipad_amd import lib as _lib


def deformable_aggregation_forward(mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights):
    """-> output [bs, num_anchors, num_embeds] (reference: deformable_aggregation.cpp:31-62)."""
    return _lib.daf_forward(mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights)


def deformable_aggregation_backward(mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights,
                                    grad_output, grad_mc_ms_feat, grad_sampling_location, grad_weights):
    """Accumulates into the three caller-allocated gradient tensors (reference:
    deformable_aggregation.cpp:86-124; its caller zero-fills them)."""
    _lib.daf_backward(mc_ms_feat, spatial_shape, scale_start_index, sampling_location, weights,
                      grad_output, grad_mc_ms_feat, grad_sampling_location, grad_weights, overwrite_loc_w=False)
