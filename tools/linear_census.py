"""Shapes of every Linear call of one steady-state training frame, then a per-shape timing of the three
GEMM kernels (forward, dX, dW) on those shapes."""
import collections, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import hipad_amd  # noqa
import torch
from hipad_amd import lib
from hipad_amd.frame import build_detector, SyntheticFrames, TrainStep
model, cfg = build_detector(stage=2)
model.train()
frames = SyntheticFrames(); step = TrainStep(model, cfg)
for _ in range(3):
    step(*frames.next())
fwd, bwd = collections.Counter(), collections.Counter()
of, ob = lib.linear_forward, lib.linear_backward
def f(x2, w, b, relu):
    fwd[(x2.shape[0], w.shape[0], x2.shape[1], bool(relu), b is not None)] += 1
    return of(x2, w, b, relu)
def b_(dy2, y_relu, x2, w, dx, dw, db):
    bwd[(x2.shape[0], w.shape[0], x2.shape[1], y_relu is not None, dx is not None, dw is not None)] += 1
    return ob(dy2, y_relu, x2, w, dx, dw, db)
lib.linear_forward, lib.linear_backward = f, b_
step(*frames.next())
torch.cuda.synchronize()
lib.linear_forward, lib.linear_backward = of, ob
print("forward calls", sum(fwd.values()), "distinct", len(fwd), "| backward calls", sum(bwd.values()))


def timeit(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


tot = collections.Counter()
print("%6s %6s %6s relu  count   fwd_us    dx_us    dw_us   frame_ms(fwd+dx+dw)" % ("M", "N", "K"))
rows = []
shapes = collections.Counter()
for (M, N, K, relu, hb), c in fwd.items():
    shapes[(M, N, K)] += c
for (M, N, K), c in sorted(shapes.items(), key=lambda kv: -kv[1]):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    dy = torch.randn(M, N, device="cuda"); dx = torch.empty_like(x); dw = torch.zeros_like(w); db = torch.zeros_like(b)
    tf = timeit(lambda: of(x, w, b, False))
    tx = timeit(lambda: ob(dy, None, x, w, dx, None, None))
    tw = timeit(lambda: ob(dy, None, x, w, None, dw, db))
    rows.append((M, N, K, c, tf, tx, tw))
    tot["fwd"] += c * tf; tot["dx"] += c * tx; tot["dw"] += c * tw
for M, N, K, c, tf, tx, tw in rows[:60]:
    print("%6d %6d %6d       %5d %8.1f %8.1f %8.1f   %8.2f" % (M, N, K, c, tf, tx, tw, c * (tf + tx + tw) / 1e3))
print("per-frame totals (ms, eager back-to-back launches): fwd %.2f dx %.2f dw %.2f" % (tot["fwd"] / 1e3, tot["dx"] / 1e3, tot["dw"] / 1e3))
