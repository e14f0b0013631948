"""Experiment: does the captured training step survive ROCm's graph packet-capture fast path when MIOpen is held to its
deterministic solvers (no memset + atomic-accumulate weight-gradient kernels)?  Prints per-step time, loss and pre-clip
gradient norm; run once per setting:

    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python tools/try_fastpath.py            # the safe path (reference trace)
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 DET=1 python tools/try_fastpath.py      # fast path, deterministic MIOpen
"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import hipad_amd  # noqa: F401
import torch
from hipad_amd import runtime_env
from hipad_amd.frame import build_detector, SyntheticFrames, GraphedTrainStep
runtime_env.graph_replay_is_safe = lambda: True   # the experiment is exactly about the unsafe setting
if os.environ.get("DET") == "1":
    torch.backends.cudnn.deterministic = True
torch.manual_seed(1234)
model, cfg = build_detector(stage=2, plan_queries=480)
model.train()
frames = SyntheticFrames(seed=0)
t = time.perf_counter()
step = GraphedTrainStep(model, cfg, frames)
torch.cuda.synchronize()
print("env", os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"), "deterministic", torch.backends.cudnn.deterministic,
      "capture %.1f s" % (time.perf_counter() - t), flush=True)
ts = []
for i in range(int(os.environ.get("NSTEPS", "30"))):
    torch.cuda.synchronize(); t = time.perf_counter()
    loss = step()
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t))
    print(i, "loss %.4f gnorm %.4e  %.2f ms" % (float(loss), float(step.inner.grad_norm), ts[-1]), flush=True)
ts.sort()
print("median ms", ts[len(ts) // 2])
