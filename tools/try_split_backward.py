"""The captured step with the backward split at the pyramid levels into two graphs (HIPAD_SPLIT_FORWARD=1
HIPAD_SPLIT_BACKWARD=1 force the multi-rank schedule on one rank): does the capture survive, and do the replays compute
what the unsplit capture computes?  Run under `timeout`: a failed capture of this kind has ended in a segfault."""
import os, sys, time, warnings
os.environ.setdefault("HIPAD_SPLIT_FORWARD", "1")
os.environ.setdefault("HIPAD_SPLIT_BACKWARD", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import hipad_amd  # noqa
import torch
from hipad_amd.frame import build_detector, SyntheticFrames, GraphedTrainStep
torch.manual_seed(1234)
model, cfg = build_detector(stage=2, plan_queries=480)
model.train()
frames = SyntheticFrames(seed=0)
t = time.perf_counter()
step = GraphedTrainStep(model, cfg, frames)
torch.cuda.synchronize()
print("capture done in %.1f s; graphs: F %s L %s E %s" % (time.perf_counter() - t, step.graph_f is not None, step.graph_l is not None,
                                                        step.graph_e is not None), flush=True)
for i in range(8):
    torch.cuda.synchronize(); t = time.perf_counter()
    loss = step()
    torch.cuda.synchronize()
    print(i, "loss %.4f  grad norm %.2f  %.2f ms" % (float(loss), float(step.inner.grad_norm), 1e3 * (time.perf_counter() - t)), flush=True)
