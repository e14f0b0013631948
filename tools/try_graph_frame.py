"""Whole training step replayed from hipGraphs: timing + loss trace (GPU box)."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import hipad_amd  # noqa: F401  HIP runtime flags before torch
import torch
from hipad_amd.frame import build_detector, SyntheticFrames, GraphedTrainStep
plan = int(sys.argv[1]) if len(sys.argv) > 1 else 480
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if os.environ.get("BENCH") == "1":
    torch.backends.cudnn.benchmark = True
if os.environ.get("IMM") == "1":
    torch.backends.miopen.immediate = True
if os.environ.get("SEED1234"):
    torch.manual_seed(1234)
model, cfg = build_detector(stage=2, plan_queries=plan)
model.train()
frames = SyntheticFrames(bs=bs)
t = time.perf_counter()
step = GraphedTrainStep(model, cfg, frames)
torch.cuda.synchronize()
print("capture done in %.1f s" % (time.perf_counter() - t), flush=True)
if os.environ.get("WITH_DAF"):
    sys.path.insert(0, ROOT)
    from bench import DafStage2
    daf = DafStage2(torch.device("cuda", 0), 0, plan)
    print("DafStage2 built", flush=True)
for i in range(int(os.environ.get("NSTEPS", "12"))):
    torch.cuda.synchronize(); t = time.perf_counter()
    loss = step()
    torch.cuda.synchronize()
    print(i, f"loss {float(loss):.4f}  {1e3*(time.perf_counter()-t):.2f} ms  mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB", flush=True)
