"""Fused BatchNorm(+add)+ReLU kernels vs the library path (MIOpen batch norm + torch add / relu) on the encoder's layer
shapes, forward + backward, each replayed from a hipGraph of 10 repetitions (GPU time per repetition, no host overhead).

    python tools/bench_bn.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import hipad_amd  # noqa
import torch
from hipad_amd import functional as HF
from projects.mmdet3d_plugin.models import image_encoder as IE

SHAPES = [(64, 128, 352, False), (64, 64, 176, False), (256, 64, 176, True), (128, 32, 88, False), (512, 32, 88, True),
          (256, 16, 44, False), (1024, 16, 44, True), (512, 8, 22, False), (2048, 8, 22, True)]
REPS = 10


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(REPS):
                fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (5 * REPS)


print("%5s %9s %4s | fused us | library us | bytes-bound us (20 B/elem @ 8 TB/s)" % ("C", "HxW", "res"))
for C, h, w, res in SHAPES:
    bn = IE.BatchNorm2d(C).cuda().train()
    bn.weight.grad, bn.bias.grad = torch.zeros_like(bn.weight), torch.zeros_like(bn.bias)
    x = torch.randn(6, C, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    r = torch.randn_like(x).requires_grad_(True) if res else None
    g = torch.randn_like(x)

    def step():
        x.grad = None
        y = bn(x, relu=True, residual=r)
        y.backward(g)

    IE.USE_FUSED_BN = True
    HF.BN_ARENA.reset(x.device, 1 << 22)
    t_f = timed(step)
    IE.USE_FUSED_BN = False
    t_l = timed(step)
    print("%5d %9s %4s | %8.1f | %10.1f | %6.1f" % (C, "%dx%d" % (h, w), res, t_f, t_l, 6 * C * h * w * 20 / 8e6))
