"""GPU time of the three Linear kernels per shape, measured inside a replayed hipGraph (no host launch floor)."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import hipad_amd  # noqa
import torch
from hipad_amd import lib
SHAPES = [(48, 256, 256, 108), (100, 256, 256, 71), (144, 256, 256, 60), (900, 256, 256, 54), (1, 256, 256, 48),
          (900, 32, 32, 42), (5400, 256, 256, 36), (6, 256, 12, 24), (900, 512, 512, 22), (481, 256, 256, 22),
          (900, 128, 128, 21), (100, 512, 256, 11), (900, 1024, 512, 6), (100, 9600, 256, 6), (6, 9600, 256, 6),
          (480, 2880, 256, 6), (6, 2880, 256, 6), (1481, 1024, 512, 6), (1481, 256, 1024, 6), (1481, 256, 512, 6)]


def graph_time(fn, reps=40):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / (2 * reps) * 1e3


tot = [0.0, 0.0, 0.0]
print("%6s %6s %6s count   fwd_us    dx_us    dw_us" % ("M", "N", "K"))
for M, N, K, c in SHAPES:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
    dy = torch.randn(M, N, device="cuda"); dx = torch.empty_like(x); dw = torch.zeros_like(w); db = torch.zeros_like(b)
    y = torch.empty(M, N, device="cuda")
    tf = graph_time(lambda: lib.linear_forward(x, w, b, True))
    tx = graph_time(lambda: lib.linear_backward(dy, y, x, w, dx, None, None))
    tw = graph_time(lambda: lib.linear_backward(dy, y, x, w, None, dw, db))
    tot[0] += c * tf; tot[1] += c * tx; tot[2] += c * tw
    print("%6d %6d %6d %5d %8.1f %8.1f %8.1f" % (M, N, K, c, tf, tx, tw), flush=True)
print("weighted totals over these shapes (ms/frame): fwd %.2f dx %.2f dw %.2f" % tuple(t / 1e3 for t in tot))
