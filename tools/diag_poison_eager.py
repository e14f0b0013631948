"""Does any kernel of the eagerly launched training step read memory it (or an earlier kernel of the step) did not write?
Before the forward and again before the backward every FREE block of the caching allocator is filled with a poison
value (NaN by default, POISON=<float> for another) and released again, so the step's torch.empty() buffers start out
poisoned.  A kernel that reads such memory shows up as non-finite gradients (NaN) or as gradients that depend on the
poison value (run with POISON=0 and POISON=1e30 and compare the printed norms)."""
import gc, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")
import hipad_amd  # noqa
import torch
from hipad_amd import functional as HF
from hipad_amd.frame import build_detector, SyntheticFrames, TrainStep
from test_graph_step_gpu import quiet

POISON = float(os.environ.get("POISON", "nan"))


def poison_free_blocks():
    torch.cuda.synchronize()
    sizes = []
    for seg in torch.cuda.memory._snapshot()["segments"]:
        for b in seg["blocks"]:
            if b["state"] == "inactive":
                sizes.append(b["size"])
    held = [torch.empty(s // 4, dtype=torch.float32, device="cuda") for s in sorted(sizes, reverse=True)]
    for t in held:
        t.fill_(POISON)
    torch.cuda.synchronize()
    n = sum(t.numel() for t in held)
    del held
    return n


torch.manual_seed(5)
model, cfg = build_detector(stage=2, plan_queries=480)
model.train(); quiet(model)
cfg["optimizer"] = dict(cfg["optimizer"], lr=0.0, weight_decay=0.0)
frames = SyntheticFrames(seed=3)
step = TrainStep(model, cfg)
names = {id(p): n for n, p in model.named_parameters()}
for k in range(int(os.environ.get("NSTEPS", "4"))):
    img, data = frames.next()
    gc.collect()
    n0 = poison_free_blocks() if k >= 1 else 0
    step.part_forward(img, data, keep_levels=True)
    step.exchange_counts()
    n1 = poison_free_blocks() if k >= 1 else 0
    loss = step.part_loss_backward()
    step.part_backward_encoder()
    step.grads.check_views()
    torch.cuda.synchronize()
    flat = step.grads.flat
    bad = [names[id(p)] for p in step.grads.params if not torch.isfinite(p.grad).all()]
    print("frame %d poison %s (%d + %d floats): loss %.5f  |grad| %.4f  non-finite params %d %s"
          % (k, POISON, n0, n1, float(loss), float(flat.double().norm()), len(bad), bad[:10]), flush=True)
    step.update()
    HF.advance_dropout_clock(img.device)
