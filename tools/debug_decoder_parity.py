"""Print the relative error of every stored decoder output vs the reference golden (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import warnings; warnings.filterwarnings("ignore")
import numpy as np, torch
from conftest import load_golden
from test_decoder import build_decoder, run_two_frames
from seeded import fill_parameters_by_name

z = load_golden("decoder_stage2")
dec = build_decoder(tuple(z["input_hw"]))
fill_parameters_by_name(dec, 4242)
dec = dec.cuda().eval()
outs = run_two_frames(dec, z, "cuda")
for step, (det, mp, ego, plan, motion, _) in enumerate(outs):
    for li in (0, 5):
        items = dict(det_cls=det["classification"][li], det_box=det["prediction"][li], det_qt=det["quality"][li],
                     map_cls=mp["classification"][li], map_pts=mp["prediction"][li], plan_cls=plan["classification"][li],
                     plan_reg=plan["prediction"][li], ego_status=ego["status"][li], motion_cls=motion["classification"][li])
        for k, t in items.items():
            ref = z[f"s{step}_{k}_{li}"]; a = t.float().cpu().numpy()
            d = np.abs(a - ref)
            print(f"s{step} L{li} {k:11s} max_rel {d.max()/np.abs(ref).max():.5f}  mean_rel {d.mean()/np.abs(ref).mean():.6f}  refmax {np.abs(ref).max():.3f}")
    same = (det["classification"][5].argmax(-1)[0].cpu().numpy() == z[f"s{step}_det_cls_5"].argmax(-1)[0]).mean()
    print("argmax agreement L5", same)

# ---- step 1: slot order of the temporal queries depends on top-k over confidences that differ in the
# 4th digit; match rows by box and compare again
det = outs[1][0]
mine_box = det["prediction"][5][0].float().cpu().numpy(); ref_box = z["s1_det_box_5"][0]
d = ((ref_box[:, None, :3] - mine_box[None, :, :3]) ** 2).sum(-1)
match = d.argmin(1)
print("matched rows unique:", len(set(match.tolist())), "of", len(match), " identity frac", (match == np.arange(len(match))).mean())
for k, t in dict(det_cls=det["classification"][5], det_box=det["prediction"][5], det_qt=det["quality"][5]).items():
    a = t[0].float().cpu().numpy()[match]; ref = z[f"s1_{k}_5"][0]
    dd = np.abs(a - ref)
    print(f"s1 L5 {k} after matching: max_rel {dd.max()/np.abs(ref).max():.5f} mean_rel {dd.mean()/np.abs(ref).mean():.6f}")
