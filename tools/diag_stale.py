"""Which tensors does the captured training step read that were allocated BEFORE the capture and freed
afterwards?  (Their addresses are baked into the graph; once the caching allocator hands the memory to
someone else, replays read garbage.)  Compares allocator snapshots around the capture and prints the
allocation stacks of every block that was live before and is free after (GPU box)."""
import gc, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import torch
from hipad_amd.frame import build_detector, SyntheticFrames, GraphedTrainStep

torch.manual_seed(1234)
model, cfg = build_detector(stage=2, plan_queries=480)
model.train()
frames = SyntheticFrames(seed=0)
snaps = {}


def live_blocks(snap):
    out = {}
    for seg in snap["segments"]:
        addr = seg["address"]
        for b in seg["blocks"]:
            if b["state"] == "active_allocated":
                out[addr] = (b["size"], b.get("frames", []), seg.get("segment_pool_id"))
            addr += b["size"]
    return out


def hook(tag):
    if tag == "before_capture":
        gc.collect()
        torch.cuda.synchronize()
    snaps[tag] = live_blocks(torch.cuda.memory._snapshot())


torch.cuda.memory._record_memory_history(max_entries=2_000_000, stacks="python")
GraphedTrainStep.debug_hook = staticmethod(hook)
step = GraphedTrainStep(model, cfg, frames)
torch.cuda.synchronize()
gc.collect()
after = live_blocks(torch.cuda.memory._snapshot())
before = snaps["before_capture"]
gone = {a: v for a, v in before.items() if a not in after}
print("live before capture: %d blocks; of those freed by now: %d" % (len(before), len(gone)))
seen = {}
for a, (size, frames_, pool) in sorted(gone.items()):
    fr = [f for f in frames_ if "/root/repo" in f["filename"] or "repo/" in f["filename"]][:6]
    key = tuple((f["filename"].split("repo/")[-1], f["line"]) for f in fr)
    seen.setdefault(key, []).append(size)
for key, sizes in sorted(seen.items(), key=lambda kv: -sum(kv[1])):
    print("%4d blocks %10d bytes  " % (len(sizes), sum(sizes)), " <- ".join("%s:%d" % k for k in key))
