"""Flat gradient of the replayed (hipGraph) training step against the eagerly launched one, frame by frame, with the
decoder's discrete choices (temporal top-k, motion-mode class) of ONE recorded eager run replayed in every other run, the
stochastic layers and the weight update off: what is left between two runs is float summation order.  Prints the relative
L2 distance of the decoder segment and of the encoder segment of the flat gradient for frames 5..9:
    eager(recorded) vs eager(replayed choices)   -- the noise floor
    eager(recorded) vs graph(replayed choices)   -- the replay
usage: python tools/graph_vs_eager_pinned.py [frames]     (tests/test_graph_step_gpu.py asserts the same)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import hipad_amd  # noqa: F401  (before torch)
import warnings
warnings.filterwarnings("ignore")
from test_graph_step_gpu import pinned_run, segment_distances

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
scope = sys.argv[2] if len(sys.argv) > 2 else "all"
first = int(sys.argv[3]) if len(sys.argv) > 3 else 5      # first frame reported for the eager runs (the graph run: 5)
import torch
enc = torch.float32 if os.environ.get("ENCODER") == "fp32" else None   # fp32 encoder: no bf16 rounding flips in the pyramid
pinned_run("eager", 2, None, scope, 0, enc)      # throw-away: MIOpen's find phase runs other solvers on a shape's first call
ref, choices = pinned_run("eager", n, None, scope, first, enc)
again, _ = pinned_run("eager", n, choices, scope, first, enc)
graph, _ = pinned_run("graph", n, choices, scope, 5, enc) if n > 5 else ([], None)
for name, other, base in (("eager vs eager", again, first), ("eager vs graph", graph, 5)):
    for k, (a, b) in enumerate(zip(ref[base - first:], other)):
        d = segment_distances(a, b)
        print("%s frame %d: loss %.4f | %.4f  decoder segment %.2e  encoder segment %.2e  norms %.1f | %.1f"
              % (name, base + k, a["loss"], b["loss"], d[0], d[1], a["norm"], b["norm"]), flush=True)

# which parameters carry the distance (frame 6, eager vs eager): squared-difference share, norms on both sides
a, b = ref[-1], again[-1]
rows = []
total = float((a["flat"] - b["flat"]).double().pow(2).sum())
for name, off, n in a["layout"]:
    da = a["flat"][off:off + n].double(); db = b["flat"][off:off + n].double()
    rows.append((float((da - db).pow(2).sum()) / total, float(da.norm()), float(db.norm()), name, n))
rows.sort(reverse=True)
print("top parameters by share of |eager - eager|^2 on the last frame:")
for share, na, nb, name, n in rows[:25]:
    print("  %6.2f%%  |a| %10.3f  |b| %10.3f  n %8d  %s" % (100 * share, na, nb, n, name))
