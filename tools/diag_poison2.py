"""Which free block of the regular pool does the captured step read?  Poison them one at a time."""
import gc, os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import torch
from hipad_amd.frame import build_detector, SyntheticFrames, GraphedTrainStep


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


torch.manual_seed(1234)
model, cfg = build_detector(stage=2, plan_queries=480)
model.train()
frames = SyntheticFrames(seed=0)
step = GraphedTrainStep(model, cfg, frames)
torch.cuda.synchronize(); gc.collect()
by_ptr = {p.data_ptr(): n for n, p in model.named_parameters()}
ordered = [by_ptr[p.data_ptr()] for p in step.inner.params]
held = grab_free()
print("free blocks:", [(hex(t.data_ptr()), t.numel() * 4) for t in held], flush=True)
for k in range(-1, len(held)):
    for j, t in enumerate(held):
        t.fill_(float("nan") if j == k else 0.0)
    torch.cuda.synchronize()
    step._feed(*frames.next())
    step.graph_a.replay()
    torch.cuda.synchronize()
    bad = [n for n, p in zip(ordered, step.inner.params) if not torch.isfinite(p.grad).all()]
    print("poisoned block", k, "" if k < 0 else (hex(held[k].data_ptr()), held[k].numel() * 4), "loss", float(step.loss),
          "bad grads:", len(bad), bad[:6], flush=True)
