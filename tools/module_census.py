"""Kernel launches and GPU time of one eager training frame's FORWARD, by decoder module type (forward hooks +
torch.profiler), to find where the launch count comes from."""
import collections, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import hipad_amd  # noqa
import torch
from torch.profiler import profile, ProfilerActivity, record_function
from hipad_amd.frame import build_detector, SyntheticFrames, TrainStep, frame_losses
model, cfg = build_detector(stage=2)
model.train()
frames = SyntheticFrames(); step = TrainStep(model, cfg)
for _ in range(3):
    step(*frames.next())
dec = model.head.onedecoder_head
groups = {}
for name, mod in dec.named_children():
    if isinstance(mod, torch.nn.ModuleList):
        if name == "layers":
            for op, m in zip(dec.operation_order, mod):
                if m is not None:
                    groups[m] = "layers." + op
        else:
            for m in mod:
                groups[m] = name
    else:
        groups[mod] = name
ctx = {}
def pre(mod, inp):
    ctx[mod] = record_function("MOD::" + groups[mod]); ctx[mod].__enter__()
def post(mod, inp, out):
    ctx.pop(mod).__exit__(None, None, None)
for m in groups:
    m.register_forward_pre_hook(pre); m.register_forward_hook(post)
torch.cuda.synchronize()
img, data = frames.next()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    with record_function("MOD::encoder"):
        fm, depths = model.extract_feat(img, True, data)
    with record_function("MOD::decoder_total"):
        outs = model.head(img, fm, data)
    with record_function("MOD::losses"):
        losses = model.head.loss(outs, data)
        total = sum(losses.values())
    with record_function("MOD::backward"):
        total.backward()
    torch.cuda.synchronize()
# attribute kernels to the innermost MOD:: range by time containment on the CPU launch side
events = prof.events()
mods = [e for e in events if e.name.startswith("MOD::")]
launch = [e for e in events if e.name in ("hipLaunchKernel", "hipExtModuleLaunchKernel", "hipMemcpyAsync", "hipMemsetAsync")]
agg = collections.defaultdict(lambda: [0, 0.0])
for e in launch:
    t = e.time_range.start
    inner = None
    for m in mods:
        if m.time_range.start <= t <= m.time_range.end and (inner is None or m.time_range.start >= inner.time_range.start):
            inner = m
    key = inner.name[5:] if inner else "(none)"
    agg[key][0] += 1
print("%-34s %8s" % ("module", "launches"))
for k, (n, _) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%-34s %8d" % (k, n))
for m in mods:
    if m.name in ("MOD::encoder", "MOD::decoder_total", "MOD::losses", "MOD::backward"):
        print(m.name, "cpu ms %.1f" % ((m.time_range.end - m.time_range.start) / 1e3))
