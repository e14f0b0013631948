"""Per-shape kernel durations of the MLP-chain kernels and of the per-layer kernels they replace: run every case a few
times eagerly under `rocprofv3 --kernel-trace` and read the durations with tools/trace_summary.py (a one-node hipGraph
holding a chain launch crashes ROCm 7.2's capture_end, so no graph here).

    rocprofv3 --kernel-trace --output-format csv -d out -- python tools/bench_chain.py [cold] ; python tools/trace_summary.py out 60
"""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import hipad_amd  # noqa
import torch
from hipad_amd import functional as HF
from hipad_amd.compat import Linear, MLPStack, Scale
from projects.mmdet3d_plugin.models.blocks import linear_relu_ln, mlp_head

COLD = len(sys.argv) > 1 and sys.argv[1] == "cold"
torch.manual_seed(0)
dev = torch.device("cuda")


def stacks():
    yield "reg 5L+2LN M=900", MLPStack(*linear_relu_ln(256, 2, 2), Linear(256, 11), Scale([1.0] * 11)), 900, 256
    yield "cls 3L+2LN M=900", MLPStack(*linear_relu_ln(256, 1, 2), Linear(256, 10)), 900, 256
    yield "reg 5L+2LN M=48", MLPStack(*linear_relu_ln(256, 2, 2), Linear(256, 12), Scale([1.0] * 12)), 48, 256
    yield "enc 2L+2LN K=3 N=128 M=900", MLPStack(*linear_relu_ln(128, 1, 2, 3)), 900, 3
    yield "cam 2L+2LN K=12 M=6", MLPStack(*linear_relu_ln(256, 1, 2, 12)), 6, 12
    yield "mlp 3L M=5400", mlp_head(256, 24), 5400, 256
    yield "one 1L M=900", MLPStack(Linear(256, 256)), 900, 256


evict = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for name, mod, M, K in stacks():
    mod = mod.to(dev)
    for p in mod.parameters():
        p.grad = torch.zeros_like(p)
    x = torch.randn(1, M, K, device=dev, requires_grad=True)
    for mode in ("chain", "layers"):
        HF.USE_CHAINS = mode == "chain"
        for rep in range(6):
            if COLD:
                evict.fill_(1)
            y = mod(x)
            y.backward(torch.ones_like(y))
            torch.cuda.synchronize()
    HF.USE_CHAINS = True
    print("ran", name, flush=True)
