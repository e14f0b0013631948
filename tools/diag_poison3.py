"""Poison the free blocks of the regular pool (HOLD=1: keep them allocated, HOLD=0: free them again) and
replay A (and B if WITH_B=1); report non-finite grads after A and after B."""
import gc, os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import torch
from hipad_amd.frame import build_detector, SyntheticFrames, GraphedTrainStep
HOLD = os.environ.get("HOLD", "0") == "1"
WITH_B = os.environ.get("WITH_B", "1") == "1"


def grab_free():
    torch.cuda.synchronize()
    snap = torch.cuda.memory._snapshot()
    sizes = []
    for seg in snap["segments"]:
        if tuple(seg.get("segment_pool_id", (0, 0))) != (0, 0):
            continue
        for b in seg["blocks"]:
            if b["state"] == "inactive":
                sizes.append(b["size"])
    return [torch.empty(s // 4, dtype=torch.float32, device="cuda") for s in sorted(sizes, reverse=True)]


if os.environ.get("IMM") == "1":
    torch.backends.miopen.immediate = True
if os.environ.get("BENCH") == "1":
    torch.backends.cudnn.benchmark = True
torch.manual_seed(1234)
model, cfg = build_detector(stage=2, plan_queries=480)
model.train()
frames = SyntheticFrames(seed=0)
step = GraphedTrainStep(model, cfg, frames)
torch.cuda.synchronize(); gc.collect()
by_ptr = {p.data_ptr(): n for n, p in model.named_parameters()}
ordered = [by_ptr[p.data_ptr()] for p in step.inner.params]
print("HOLD", HOLD, "WITH_B", WITH_B, flush=True)
for k in range(int(os.environ.get("NSTEPS", "8"))):
    held = grab_free()
    for t in held:
        t.fill_(float(os.environ.get("POISON", "nan")))
    torch.cuda.synchronize()
    if not HOLD:
        del held
    step._feed(*frames.next())
    step.graph_f.replay(); (step.graph_l.replay() if step.graph_l is not None else None)
    torch.cuda.synchronize()
    flat = step.inner.grads.flat
    nbad_a = int((~torch.isfinite(flat)).sum())
    print("   pre-clip gnorm %.4e max|g| %.4e" % (float(flat.double().norm()), float(flat.abs().max())), flush=True)
    nbad_b = -1
    if WITH_B:
        step.graph_b.replay()
        torch.cuda.synchronize()
        nbad_b = int((~torch.isfinite(flat)).sum())
    badp = sum(int(not torch.isfinite(p).all()) for p in model.parameters())
    print("step", k, "loss", float(step.loss), "non-finite grad elements after A:", nbad_a, "after B:", nbad_b, "bad params", badp, flush=True)
    if nbad_a or nbad_b > 0:
        bad = [n for n, p in zip(ordered, step.inner.params) if not torch.isfinite(p.grad).all()]
        print("   ", len(bad), bad[:8])
        break
