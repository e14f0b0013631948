"""How long does the HOST spend in hipGraphLaunch for the ~9.6k-node training-step graph, and does
back-to-back replay pipeline?"""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import hipad_amd  # noqa: F401  HIP runtime flags before torch
import torch
from hipad_amd.frame import build_detector, SyntheticFrames, GraphedTrainStep
model, cfg = build_detector(stage=2)
model.train()
frames = SyntheticFrames()
step = GraphedTrainStep(model, cfg, frames)
torch.cuda.synchronize()
for mode in ("sync_each", "no_sync", "sync_each"):
    t0 = time.perf_counter(); host = 0.0
    for i in range(10):
        a = time.perf_counter()
        step()
        host += time.perf_counter() - a
        if mode == "sync_each":
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(mode, "ms/step %.1f" % ((time.perf_counter() - t0) * 100), "host ms in step() %.1f" % (host * 100), flush=True)
# split host time
torch.cuda.synchronize()
a = time.perf_counter(); step._feed(*frames.next()); b = time.perf_counter(); step.graph_a.replay(); c = time.perf_counter(); step.graph_b.replay(); d = time.perf_counter()
torch.cuda.synchronize(); e = time.perf_counter()
print("feed %.2f ms, launch A %.2f ms, launch B %.2f ms, drain %.2f ms" % ((b-a)*1e3, (c-b)*1e3, (d-c)*1e3, (e-d)*1e3))
