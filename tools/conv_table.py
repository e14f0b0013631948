"""Per-shape table of the image encoder's convolutions (ResNet50 + FPN on 6 x 256 x 704, bf16 channels-last, the
library's kernels as the training step calls them): GFLOP, microseconds and share of the 2.5 PFLOP/s dense bf16 MFMA peak
for forward / input gradient / weight gradient, one row per distinct (Cin, Cout, kernel, stride, H, W) with its count.
HIP-event timing of 20 calls replayed from a hipGraph (no host launch cost) after MIOpen's exhaustive find
(torch.backends.cudnn.benchmark); each direction through aten.convolution_backward's output mask.  GPU box.

    python tools/conv_table.py > profiles/r03_encoder_conv_table.txt
"""
import collections, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import hipad_amd  # noqa
import torch
import torch.nn.functional as F
from hipad_amd.frame import build_detector

PEAK = 2.5e15
torch.manual_seed(0)
model, _ = build_detector(stage=2, plan_queries=48)
model.train(); model.use_grid_mask = False
shapes = collections.OrderedDict()


def hook(mod, inp, out):
    x = inp[0]
    key = (mod.in_channels, mod.out_channels, mod.kernel_size, mod.stride, mod.padding, tuple(x.shape))
    shapes[key] = shapes.get(key, 0) + 1


hs = [m.register_forward_hook(hook) for m in list(model.img_backbone.modules()) + list(model.img_neck.modules())
      if isinstance(m, torch.nn.Conv2d)]
with torch.no_grad():
    model.extract_feat(torch.randn(1, 6, 3, 256, 704, device="cuda"), False, {})
for h in hs:
    h.remove()


def timed(fn, reps=20):
    """us per call, the calls replayed from a hipGraph (an eager call of a 15-us convolution is bound by ~70 us of host
    work per call and says nothing about the kernel)."""
    for _ in range(3):
        fn()                                   # MIOpen's find on the first call
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3      # us


rows, tot = [], dict(fwd=0.0, dx=0.0, dw=0.0, gflop=0.0)
for (cin, cout, k, stride, pad, xs), count in shapes.items():
    x = torch.randn(xs, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.randn(cout, cin, *k, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    x, w = x.detach(), w.detach()
    with torch.no_grad():
        y = F.conv2d(x, w, None, stride, pad)
        gy = torch.randn_like(y)
        gflop = 2.0 * y.numel() * cin * k[0] * k[1] / 1e9
        bwd = torch.ops.aten.convolution_backward
        args = (gy, x, w, None, list(stride), list(pad), [1, 1], False, [0, 0], 1)
        t_f = timed(lambda: F.conv2d(x, w, None, stride, pad))
        t_dx = timed(lambda: bwd(*args, [True, False, False])) if cin > 3 else 0.0
        t_dw = timed(lambda: bwd(*args, [False, True, False]))
    rows.append((count * (t_f + t_dx + t_dw), count, cin, cout, k[0], stride[0], xs[2], xs[3], gflop, t_f, t_dx, t_dw))
    tot["fwd"] += count * t_f; tot["dx"] += count * t_dx; tot["dw"] += count * t_dw; tot["gflop"] += count * gflop
rows.sort(reverse=True)
pct = lambda g, us: 100.0 * g * 1e9 / (us * 1e-6) / PEAK if us > 0 else 0.0  # noqa: E731
print("count  Cin  Cout  k s   H   W    GFLOP |   fwd us  %peak |    dX us  %peak |    dW us  %peak | us/frame (all three, x count)")
for total, count, cin, cout, k, s, H, W, g, tf, tdx, tdw in rows:
    print("%5d %4d %5d  %d %d %3d %3d %8.2f | %8.1f %6.2f | %8.1f %6.2f | %8.1f %6.2f | %9.1f"
          % (count, cin, cout, k, s, H, W, g, tf, pct(g, tf), tdx, pct(g, tdx), tdw, pct(g, tdw), total))
print("per frame: %.1f GFLOP forward (x3 with both gradients); forward %.0f us, input gradients %.0f us, weight gradients %.0f us"
      % (tot["gflop"], tot["fwd"], tot["dx"], tot["dw"]))
print("whole encoder convolutions: %.1f TFLOP/s = %.2f %% of 2.5 PFLOP/s"
      % (3 * tot["gflop"] * 1e9 / ((tot["fwd"] + tot["dx"] + tot["dw"]) * 1e-6) / 1e12,
         100 * 3 * tot["gflop"] * 1e9 / ((tot["fwd"] + tot["dx"] + tot["dw"]) * 1e-6) / PEAK))
