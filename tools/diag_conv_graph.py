"""Standalone: weight gradient of the layer4.0.downsample conv (1x1 stride 2, 1024->2048, bf16 channels-last,
input 6x1024x16x44) eager vs captured, with the regular pool's free blocks poisoned."""
import os, sys, warnings
import torch
import torch.nn.functional as F
warnings.filterwarnings("ignore")
torch.manual_seed(0)
IMM = os.environ.get("IMM", "0") == "1"
BENCH = os.environ.get("BENCH", "0") == "1"
FP32 = os.environ.get("FP32", "0") == "1"
torch.backends.miopen.immediate = IMM
torch.backends.cudnn.benchmark = BENCH
print("immediate", IMM, "benchmark", BENCH, "fp32", FP32, flush=True)
dev = "cuda"
x = torch.randn(6, 1024, 16, 44, device=dev).to(memory_format=torch.channels_last)
w = torch.randn(2048, 1024, 1, 1, device=dev, requires_grad=True) * 0.02
w = w.detach().requires_grad_(True)
gy = torch.randn(6, 2048, 8, 22, device=dev).to(memory_format=torch.channels_last)


def run():
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not FP32):
        y = F.conv2d(x, w, None, stride=2)
    (g,) = torch.autograd.grad(y, w, gy.to(y.dtype))
    return g


ref = torch.autograd.grad(F.conv2d(x.double(), w.double(), None, stride=2), w, gy.double())[0]
for _ in range(3):
    g_eager = run()
print("eager rel err", float((g_eager.double() - ref).abs().max() / ref.abs().max()))
# junk in the allocator: allocate + free a spread of sizes filled with NaN
junk = [torch.full((n,), float("nan"), device=dev) for n in (1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22, 3 << 20)]
torch.cuda.synchronize()
del junk
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    run()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    g_static = run()
for i in range(3):
    junk = [torch.full((n,), float("nan"), device=dev) for n in (1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22, 3 << 20)]
    torch.cuda.synchronize()
    del junk
    graph.replay()
    torch.cuda.synchronize()
    print("replay", i, "non-finite", int((~torch.isfinite(g_static)).sum()),
          "rel err", float((g_static.double() - ref).abs().nan_to_num(1e9).max() / ref.abs().max()))
