"""Prints (loss, pre-clip gradient norm) of frames 5.. for the eagerly launched and the replayed training step, each
run twice from the same seeds with every stochastic layer switched off, to separate run-to-run noise from a
replay defect.  usage: python tools/graph_vs_eager.py [frames]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import hipad_amd  # noqa: F401  (before torch)
import warnings
warnings.filterwarnings("ignore")
from test_graph_step_gpu import run

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for mode in ("eager", "eager", "graph", "graph"):
    tr = run(mode, n)
    tr = tr[5:] if mode == "eager" else tr
    print(mode, " ".join("%.2f/%.0f" % t for t in tr), flush=True)
