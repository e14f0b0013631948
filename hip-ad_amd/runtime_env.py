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
    """True when the process environment has the settings captured training steps need."""
    return all(os.environ.get(k) == v for k, v in REQUIRED.items())
