"""Which parameters does autograd's AccumulateGrad still spend kernels on?  Tensor hooks count the gradients that arrive
THROUGH autograd per parameter in one backward (gradients our kernels add in place never pass a hook); a parameter that
is not 'loose' (its .grad is the preset flat view) pays one add kernel per arrival, a loose one pays for every arrival
after the first."""
import collections, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import hipad_amd  # noqa
import torch
from hipad_amd.frame import build_detector, SyntheticFrames, TrainStep
model, cfg = build_detector(stage=2)
model.train()
frames = SyntheticFrames(); step = TrainStep(model, cfg)
for _ in range(3):
    step(*frames.next())
names = {id(p): n for n, p in model.named_parameters()}
loose = {id(p) for p, _ in step.grads._loose}
arrivals = collections.Counter()
for p in model.parameters():
    if p.requires_grad:
        # the engine calls tensor hooks with None for the inputs a Function returned no gradient for: not an arrival
        p.register_hook(lambda g, p=p: arrivals.__setitem__(id(p), arrivals[id(p)] + (g is not None)))
step(*frames.next())
torch.cuda.synchronize()
cost = collections.Counter()
shapes = {id(p): tuple(p.shape) for p in model.parameters()}
for pid, n in arrivals.items():
    kernels = n - 1 if pid in loose else n
    if kernels > 0:
        key = ".".join(("*" if s.isdigit() else s) for s in names[pid].split("."))
        cost[(key, "loose" if pid in loose else "preset")] += kernels
print("parameters with autograd arrivals:", sum(1 for n in arrivals.values() if n), "loose:", len(loose), "of", len(names))
print("accumulation kernels per step:", sum(cost.values()))
for (k, kind), n in cost.most_common(40):
    print("%5d  %-7s %s" % (n, kind, k))
