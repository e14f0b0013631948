"""For every FREE block of the regular allocator pool after the capture: who allocated it last?"""
import gc, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import torch
from hipad_amd.frame import build_detector, SyntheticFrames, GraphedTrainStep

torch.manual_seed(1234)
model, cfg = build_detector(stage=2, plan_queries=480)
model.train()
frames = SyntheticFrames(seed=0)
torch.cuda.memory._record_memory_history(max_entries=4_000_000, stacks="python")
mark = {}
def hook(tag):
    torch.cuda.synchronize()
    mark[tag] = len(torch.cuda.memory._snapshot()["device_traces"][0])
GraphedTrainStep.debug_hook = staticmethod(hook)
step = GraphedTrainStep(model, cfg, frames)
torch.cuda.synchronize(); gc.collect()
snap = torch.cuda.memory._snapshot()
free = []
for seg in snap["segments"]:
    if tuple(seg.get("segment_pool_id", (0, 0))) != (0, 0):
        continue
    addr = seg["address"]
    for b in seg["blocks"]:
        if b["state"] == "inactive":
            free.append((addr, b["size"]))
        addr += b["size"]
trace = snap["device_traces"][0]
print("trace entries", len(trace), "marks", mark, "free blocks", len(free))
def site(frames_):
    fr = [f for f in frames_ if "repo/" in f["filename"]][:7]
    return " <- ".join("%s:%d" % (f["filename"].split("repo/")[-1], f["line"]) for f in fr)
for addr, size in free:
    print("FREE block %#x size %d" % (addr, size))
    hits = 0
    for idx in range(len(trace) - 1, -1, -1):
        e = trace[idx]
        if e["action"] in ("alloc", "free_completed") and addr <= e["addr"] < addr + size:
            phase = "capture" if idx >= mark.get("before_capture", 0) else "warmup"
            print("    [%d %s] %s size %d  %s" % (idx, phase, e["action"], e["size"], site(e.get("frames", []))))
            hits += 1
            if hits >= 4:
                break
