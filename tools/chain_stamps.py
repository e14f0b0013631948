"""In-kernel phase stamps of one chain_fwd_kernel launch (workgroup 0, thread 0): cycles and ns between phase boundaries."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")
import hipad_amd  # noqa
import torch
from hipad_amd import lib
from hipad_amd.compat import Linear, MLPStack, Scale
from projects.mmdet3d_plugin.models.blocks import linear_relu_ln
dev = torch.device("cuda")
buf = torch.zeros(512, dtype=torch.int64, device=dev)
L = lib.load()
for name, mod, M, K in [("reg 5L M=48", MLPStack(*linear_relu_ln(256, 2, 2), Linear(256, 12), Scale([1.0] * 12)), 48, 256),
                        ("reg 5L M=900", MLPStack(*linear_relu_ln(256, 2, 2), Linear(256, 12), Scale([1.0] * 12)), 900, 256),
                        ("cam 2L K=12 M=6", MLPStack(*linear_relu_ln(256, 1, 2, 12)), 6, 12)]:
    mod = mod.to(dev)
    x = torch.randn(1, M, K, device=dev, requires_grad=True)
    for rep in range(3):
        buf.zero_()
        L.hipad_chain_debug_stamps(buf.data_ptr())
        y = mod(x)
        torch.cuda.synchronize()
        L.hipad_chain_debug_stamps(None)
    st = buf.cpu().view(-1, 2)
    n = int((st[:, 0] != 0).sum())
    print(name, "stamps", n)
    for i in range(1, n):
        dc, dr = int(st[i, 0] - st[i - 1, 0]), int(st[i, 1] - st[i - 1, 1])
        print(f"   {i:3d}: {dc:8d} cycles  {dr * 10:7d} ns")
    print("   total %d cycles, %d ns" % (int(st[n - 1, 0] - st[0, 0]), int(st[n - 1, 1] - st[0, 1]) * 10))
