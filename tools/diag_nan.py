"""Does the synthetic training step stay finite?  Eager steps with the bench's seed; prints the loss,
the gradient norm and the largest magnitude of every head output per step (GPU box)."""
import os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import torch
from hipad_amd.frame import build_detector, SyntheticFrames, TrainStep, surrogate_objective, DECODER_DTYPE
import hipad_amd.functional as HF

torch.manual_seed(int(os.environ.get("SEED", "1234")))
model, cfg = build_detector(stage=2, plan_queries=480)
model.train()
frames = SyntheticFrames(seed=0)
step = TrainStep(model, cfg)
names = ("det", "map", "ego", "plan", "motion")
for i in range(int(os.environ.get("NSTEPS", "10"))):
    img, data = frames.next()
    step.grads.zero()
    fm, depths = model.extract_feat(img, True, data)
    outs = model.head(img, fm, data)
    loss = surrogate_objective(outs, depths)
    loss.backward()
    gn = float(torch.linalg.vector_norm(step.grads.flat.float())) if hasattr(step.grads, "flat") else float("nan")
    mags = {}
    for nm, out in zip(names, outs[:5]):
        for key in ("classification", "prediction", "quality", "status"):
            ts = [t for t in (out.get(key, []) or []) if t is not None]
            if ts:
                mags[f"{nm}.{key}"] = max(float(t.float().abs().max()) for t in ts)
    fmax = float(fm[0].float().abs().max())
    print(i, f"loss {float(loss):.4f} gnorm {gn:.3e} feat {fmax:.3e}", {k: f"{v:.3g}" for k, v in mags.items()}, flush=True)
    bad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    if bad:
        print("  non-finite grads:", len(bad), bad[:8], flush=True)
    step.update()
    HF.advance_dropout_clock(img.device)
    badp = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    if badp:
        print("  non-finite params:", len(badp), badp[:8], flush=True)
        break
