"""Frame-0 forward of the training model twice in one process (same seeds, stochastic layers off): where do two runs first
differ?  Prints the relative difference of the pyramid and of the decoder state after every op of the program."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
warnings.filterwarnings("ignore")
import hipad_amd  # noqa
import torch
from hipad_amd.frame import build_detector, SyntheticFrames
from test_graph_step_gpu import quiet

enc = torch.float32 if os.environ.get("ENCODER") == "fp32" else None
runs = []
for r in range(2):
    torch.manual_seed(5)
    model, cfg = build_detector(stage=2, plan_queries=480, encoder_dtype=enc)
    model.train(); quiet(model)
    dec = model.head.onedecoder_head
    dec.with_instance_id = False
    img, data = SyntheticFrames(seed=3).next()
    log = []
    dec._probe = lambda slot, op, state: log.append((slot, op, {k: v.detach().float().clone() for k, v in state.items()}))
    fm, depths = model.extract_feat(img, True, data)
    outs = model.head(img, fm, data)
    torch.cuda.synchronize()
    runs.append((fm[0].detach().float().clone(), log, [d.detach().float().clone() for d in depths]))


def rel(a, b):
    return float((a - b).norm() / a.norm().clamp_min(1e-30))


(fa, la, da), (fb, lb, db) = runs
print("pyramid: rel L2 %.3e, max abs %.3e, equal %s" % (rel(fa, fb), float((fa - fb).abs().max()), bool(torch.equal(fa, fb))))
for (sa, opa, sta), (sb, opb, stb) in zip(la, lb):
    worst = max(((rel(sta[k], stb[k]), k) for k in sta if sta[k].dtype.is_floating_point and sta[k].shape == stb[k].shape), default=(0.0, "-"))
    print("slot %3d %-12s worst rel L2 %.3e  (%s)" % (sa, opa, worst[0], worst[1]))
