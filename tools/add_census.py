"""Where do the small torch kernels of one eagerly launched training step come from?  torch.profiler with Python stacks:
every add / cat / copy / fill / sub / mul that launches a kernel is attributed to
  forward:   the innermost frame of THIS repo on its Python stack,
  backward:  the autograd node under whose evaluate_function it runs (an `aten::add` directly under a node is the
             engine joining two gradient arrivals of one tensor).
GPU box.  python tools/add_census.py [ops ...]
"""
import collections, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")
import hipad_amd  # noqa
import torch
from torch.profiler import profile, ProfilerActivity
from hipad_amd.frame import build_detector, SyntheticFrames, TrainStep

OPS = set(sys.argv[1:]) or {"aten::add", "aten::add_", "aten::cat", "aten::copy_", "aten::fill_", "aten::zero_", "aten::sub",
                            "aten::mul", "aten::sum", "aten::index_select", "aten::gather", "aten::where", "aten::clone"}
model, cfg = build_detector(stage=2, plan_queries=480)
model.train()
frames = SyntheticFrames(seed=3)
step = TrainStep(model, cfg)
for _ in range(3):
    step(*frames.next())
torch.cuda.synchronize()
img, data = frames.next()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(img, data)
    torch.cuda.synchronize()

SHOWN = []
sites = collections.Counter()
kernels = collections.Counter()
for e in prof.events():
    if e.name not in OPS:
        continue
    nk = len(e.kernels)
    if nk == 0:
        continue
    # skip ops nested inside another counted op (aten::add_ inside aten::add ...)
    p, nested, node = e.cpu_parent, False, None
    while p is not None:
        if p.name in OPS and len(p.kernels) >= nk:
            nested = True
        if p.name.startswith("autograd::engine::evaluate_function: "):
            node = p.name.split(": ", 1)[1]
        p = p.cpu_parent
    if nested:
        continue
    if node is not None:
        direct = e.cpu_parent is not None and e.cpu_parent.name.startswith("autograd::engine::evaluate_function")
        site = "bwd %s%s" % (node, " (gradient join)" if direct else "")
    else:
        site = "fwd ?"
        for fr in e.stack or ():
            if ("hipad_amd/" in fr or "hip-ad_amd/" in fr or "projects/" in fr) and "/tools/" not in fr:
                site = "fwd " + fr.split("/repo/")[-1].replace(ROOT + "/", "").strip()
                break
        if site == "fwd ?" and not SHOWN and e.stack:
            SHOWN.append(1)
            print("(sample stack of an unattributed op:", list(e.stack)[:12], ")")
    shape = ""
    sites[(e.name, site)] += 1
    kernels[(e.name, site)] += nk
print("kernel-launching ops of one eager step by origin (ops, kernels):")
tot = 0
for (name, site), n in sorted(sites.items(), key=lambda kv: -kernels[kv[0]]):
    tot += kernels[(name, site)]
    print("%5d %5d  %-16s %s" % (n, kernels[(name, site)], name, site[:150]))
print("total kernels attributed:", tot)
