"""Graph-replayed training steps with a finiteness check after every step; on the first non-finite
value, says which parameter gradients are affected (GPU box)."""
import collections, os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import torch
from hipad_amd.frame import build_detector, SyntheticFrames, GraphedTrainStep

torch.manual_seed(int(os.environ.get("SEED", "1234")))
model, cfg = build_detector(stage=2, plan_queries=480)
model.train()
frames = SyntheticFrames(seed=0)
step = GraphedTrainStep(model, cfg, frames)
torch.cuda.synchronize()
names = [n for n, p in model.named_parameters() if p.requires_grad]
by_ptr = {p.data_ptr(): n for n, p in model.named_parameters()}
ordered = [by_ptr[p.data_ptr()] for p in step.inner.params]
for i in range(int(os.environ.get("NSTEPS", "60"))):
    t = time.perf_counter()
    torch.cuda.synchronize()
    step._feed(*frames.next())
    step.graph_f.replay(); (step.graph_l.replay() if step.graph_l is not None else None)
    torch.cuda.synchronize()
    flat = step.inner.grads.flat
    pre_bad = [n for n, p in zip(ordered, step.inner.params) if not torch.isfinite(p.grad).all()]
    if pre_bad:
        print("  BEFORE clip: %d params with non-finite grads, e.g." % len(pre_bad), pre_bad[:12], flush=True)
    from projects.mmdet3d_plugin.ops import deformable_aggregation as DA
    if DA.CROSS_CHECK_LOG:
        recs = DA.CROSS_CHECK_LOG[-24:]
        vals = torch.stack([r for _, r in recs]).cpu()
        worst = float((vals[:, 0] / vals[:, 1].clamp(min=1e-20)).max())
        flag = bool((vals[:, 2:] > 0).any()) or worst > 1e-3 or not bool(torch.isfinite(vals).all())
        if flag or i == 0:
            for (shape, _), v in zip(recs, vals):
                print("   call", shape[1:3], "max|s-a| %.3e max|a| %.3e nonfinite sorted %d loc %d w %d gout %d" % tuple(v.tolist()), flush=True)
    step.graph_b.replay()
    loss = step.loss
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t)
    ok = bool(torch.isfinite(loss)) and bool(torch.isfinite(flat).all())
    print(i, f"loss {float(loss):.4f} {ms:.1f} ms gnorm(after clip) {float(flat.norm()):.3e}", "" if ok else "NON-FINITE", flush=True)
    if not ok:
        groups = collections.Counter()
        total = collections.Counter()
        first = []
        for n, p in zip(ordered, step.inner.params):
            key = ".".join(n.split(".")[:3])
            total[key] += 1
            if not torch.isfinite(p.grad).all():
                groups[key] += 1
                if len(first) < 10:
                    first.append(n)
        print("  params with non-finite grad: %d of %d" % (sum(groups.values()), sum(total.values())))
        for k, v in sorted(groups.items()):
            print("   ", k, v, "/", total[k])
        print("  e.g.", first)
        badp = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
        print("  non-finite params:", len(badp), badp[:6], flush=True)
        break
