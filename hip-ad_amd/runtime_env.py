"""Process environment the HIP runtime must see BEFORE it initialises (i.e. before ``import torch``).

DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
    ROCm 7.2's hipGraph "packet capture" fast path (AQL packets of kernel nodes pre-built at
    instantiation) replays graphs that mix kernel nodes with memset / blit nodes incorrectly on gfx950:
    MIOpen's split-K weight-gradient and bias-gradient solvers (hipMemsetAsync + atomically accumulating
    kernel) come back with garbage (1e27..1e37) on replays although the very same graph is right on its
    first replay and the same calls are right when launched eagerly.  Measured with
    tools/diag_encoder_graph.py (captured encoder fwd+bwd vs eager: 1e27 relative error with the fast
    path, noise level without), tools/diag_poison3.py (whole training step with the allocator's free
    blocks filled with NaN: pre-clip gradient norm 1e37 / NaN with the fast path, 2.3e3 without) and
    tools/diag_conv_graph.py.  With the fast path off, hipGraphLaunch enqueues the nodes through the
    normal dispatch path: correct, and still free of the Python / dispatcher cost of an eager step.

``apply()`` only sets defaults; an explicit user setting wins.  It returns False when torch was imported
before (the runtime may have read its flags already), so callers can say so.
"""
import os
import sys

REQUIRED = {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"}


def apply():
    early = "torch" not in sys.modules
    for k, v in REQUIRED.items():
        os.environ.setdefault(k, v)
    return early


def graph_replay_is_safe():
    """True when the HIP runtime of this process is known to have started with the settings captured steps need:
    the variables hold the required values AND either the process was started with them (hipad_amd.FLAGS_PRESET) or
    ``hipad_amd`` was imported before torch (so they were set before the runtime could read them).  A caller that
    imported torch first and only then hipad_amd has the right values in os.environ but possibly not in the runtime:
    refused (the replays would return the garbage gradients described above)."""
    if not all(os.environ.get(k) == v for k, v in REQUIRED.items()):
        return False
    import hipad_amd
    return bool(hipad_amd.FLAGS_PRESET or not hipad_amd.TORCH_IMPORTED_FIRST)


def why_unsafe():
    import hipad_amd
    bad = {k: os.environ.get(k) for k, v in REQUIRED.items() if os.environ.get(k) != v}
    if bad:
        return "environment has %r, needs %r" % (bad, REQUIRED)
    if hipad_amd.TORCH_IMPORTED_FIRST and not hipad_amd.FLAGS_PRESET:
        return ("torch was imported before hipad_amd, so the HIP runtime may have started without %r; import hipad_amd "
                "first or export the variable before starting the process" % (REQUIRED,))
    return ""
