"""Sampling-weights softmax (hipad_weights_softmax_forward / _backward) on the three stage-2 call shapes, 20 forward +
backward calls each.  Run under the kernel trace and summarise per grid (a call costs more host time than its kernels
run, so host-side timing says nothing here):

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/wsm -- python3 $REPO/tools/bench_wsm.py
    python tools/trace_summary.py /tmp/wsm 40
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import hipad_amd  # noqa
import torch
from hipad_amd import functional as HF

from hipad_amd import lib as _lib
split = int(os.environ.get("WSM_SPLIT", "0"))
_lib.load().hipad_weights_softmax_set_split(split)
print("workgroups per anchor:", split or "automatic", flush=True)
g = torch.Generator().manual_seed(0)
for name, A, P in (("det", 900, 13), ("map", 100, 300), ("plan", 480, 90)):
    bs, cams, L, G = 1, 6, 4, 8
    n = L * P * G
    u = torch.randn(bs, A, n, generator=g).cuda().requires_grad_(True)
    v = torch.randn(bs, cams, n, generator=g).cuda().requires_grad_(True)
    keep = ((torch.rand(bs, A, cams, P, generator=g) > 0.1).float() / 0.9).cuda()
    gw = torch.randn(bs, A, P, cams, L, G, generator=g).cuda()
    for _ in range(20):
        w = HF.sampling_weights(u, v, keep, L, P, G)
        torch.autograd.grad(w, (u, v), gw)
    torch.cuda.synchronize()
    print("%-5s A %4d P %3d: grid %d anchors, %.1f MB of weights" % (name, A, P, bs * A, 4 * bs * A * cams * n / 1e6), flush=True)
