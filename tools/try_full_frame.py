"""First contact of the whole model with the GPU: a few training frames, timing and memory."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import hipad_amd  # noqa: F401  HIP runtime flags before torch
import torch
from hipad_amd.frame import build_detector, SyntheticFrames, TrainStep
plan = int(sys.argv[1]) if len(sys.argv) > 1 else 480
model, cfg = build_detector(stage=2, plan_queries=plan)
model.train()
frames = SyntheticFrames()
step = TrainStep(model, cfg)
for i in range(8):
    torch.cuda.synchronize(); t = time.perf_counter()
    img, data = frames.next()
    loss = step(img, data)
    torch.cuda.synchronize()
    print(i, f"loss {float(loss):.4f}  {1e3*(time.perf_counter()-t):.1f} ms  mem {torch.cuda.max_memory_allocated()/2**30:.2f} GiB", flush=True)
unused = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
print("params without grad:", len(unused), unused[:12])
