"""Eager forward vs replayed inference graph on the same frame sequence: per-output difference statistics."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import hipad_amd  # noqa
import torch
from hipad_amd.frame import GraphedInference, SyntheticFrames, build_detector
heads = []
for mode in ("eager", "eager2", "graph"):
    torch.manual_seed(11)
    model, _ = build_detector(stage=2, plan_queries=480)
    model.eval()
    frames = SyntheticFrames(seed=2)
    with torch.no_grad():
        if mode.startswith("eager"):
            model.head.onedecoder_head.with_instance_id = False
            for _ in range(6):
                img, data = frames.next()
                outs = model.head(img, model.extract_feat(img, False, data), data)
        else:
            step = GraphedInference(model, frames)
            step()
            outs = step.outs
    det, mp, ego, plan, motion, _ = outs
    heads.append([t.clone() for t in (det["classification"][-1], det["prediction"][-1], mp["prediction"][-1],
                                      plan["classification"][-1], plan["prediction"][-1], ego["status"][-1])])
names = ["det_cls", "det_box", "map_pts", "plan_cls", "plan_reg", "ego_status"]
for tag, (a, b) in (("eager vs eager", (heads[0], heads[1])), ("eager vs graph", (heads[0], heads[2]))):
    print(tag)
    for n, x, y in zip(names, a, b):
        d = (x.double() - y.double()).abs()
        print("  %-10s max %.3e median %.3e  scale %.3e  frac>1e-2*scale %.4f" % (
            n, float(d.max()), float(d.median()), float(x.abs().max()), float((d > 1e-2 * x.abs().max()).float().mean())))
