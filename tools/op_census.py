"""Count aten ops / kernel launches of one training frame, grouped by top-level module (torch.profiler)."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import torch
from torch.profiler import profile, ProfilerActivity, record_function
from hipad_amd.frame import build_detector, SyntheticFrames, TrainStep
model, cfg = build_detector(stage=2)
model.train()
frames = SyntheticFrames(); step = TrainStep(model, cfg)
for _ in range(3):
    step(*frames.next())
torch.cuda.synchronize()
img, data = frames.next()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step.grads.zero()
    with record_function("ENCODER_FWD"):
        fm, depths = model.extract_feat(img, True, data)
    with record_function("DECODER_FWD"):
        outs = model.head(img, fm, data)
    with record_function("LOSS"):
        from hipad_amd.frame import surrogate_objective
        loss = surrogate_objective(outs, depths)
    with record_function("BACKWARD"):
        loss.backward()
    with record_function("OPT"):
        step.update()
    torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.count)
print("%-60s %8s %12s %12s" % ("op", "count", "cpu_ms", "cuda_ms"))
for e in rows[:45]:
    print("%-60s %8d %12.2f %12.2f" % (e.key[:60], e.count, e.cpu_time_total / 1e3, e.device_time_total / 1e3))
for name in ("ENCODER_FWD", "DECODER_FWD", "LOSS", "BACKWARD", "OPT"):
    e = [x for x in ka if x.key == name][0]
    print(name, "cpu_ms %.1f" % (e.cpu_time_total / 1e3), "cuda_ms %.1f" % (e.device_time_total / 1e3))
