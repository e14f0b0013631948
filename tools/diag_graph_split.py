"""Which small hipGraph captures crash ROCm 7.2's capture_end (segmentation fault)?  Each case runs in its own process.

    python tools/diag_graph_split.py            # runs all cases, prints OK / CRASH(rc) per case
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["chain_only", "dummy_then_chain", "chain_then_dummy", "bwd_only_conv", "fwd_graph_then_bwd_graph_conv", "dummy_then_bwd_conv",
         "bwd_only_linear", "bwd_only_conv_nobench"]


def case(name):
    sys.path.insert(0, ROOT)
    import hipad_amd  # noqa
    import torch
    from hipad_amd.compat import Linear, MLPStack
    dev = torch.device("cuda")
    if name != "bwd_only_conv_nobench":
        torch.backends.cudnn.benchmark = True
    g = torch.cuda.CUDAGraph()
    if name in ("chain_only", "dummy_then_chain", "chain_then_dummy"):
        mod = MLPStack(Linear(256, 256), Linear(256, 256)).to(dev)
        x = torch.randn(64, 256, device=dev)
        with torch.no_grad():
            mod(x); torch.cuda.synchronize()
            with torch.cuda.graph(g):
                if name == "dummy_then_chain":
                    y0 = x + 1
                y = mod(x)
                if name == "chain_then_dummy":
                    y1 = y + 1
        g.replay(); torch.cuda.synchronize()
        return
    if "conv" in name:
        net = torch.nn.Sequential(torch.nn.Conv2d(16, 32, 3, padding=1), torch.nn.BatchNorm2d(32), torch.nn.ReLU(),
                                  torch.nn.Conv2d(32, 32, 3, padding=1)).to(dev).to(memory_format=torch.channels_last)
        x = torch.randn(6, 16, 64, 176, device=dev).contiguous(memory_format=torch.channels_last)
    else:
        net = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.ReLU(), torch.nn.Linear(64, 8)).to(dev)
        x = torch.randn(32, 64, device=dev)
    for _ in range(3):   # warm-up (MIOpen find) on a side stream like torch recommends
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = net(x)
            y.float().square().mean().backward()
        torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    net.zero_grad(set_to_none=True)
    if name == "fwd_graph_then_bwd_graph_conv":
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = net(x)
            gy = torch.ones_like(y)
        with torch.cuda.graph(g, pool=g1.pool()):
            torch.autograd.backward([y], [gy])
        g1.replay(); g.replay(); torch.cuda.synchronize()
        return
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = net(x)
    gy = torch.ones_like(y)
    with torch.cuda.graph(g):
        if name == "dummy_then_bwd_conv":
            z = gy + 1
        torch.autograd.backward([y], [gy])
    g.replay(); torch.cuda.synchronize()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        case(sys.argv[1])
        print("finished", sys.argv[1])
    else:
        for c in CASES:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True, timeout=300)
            print(f"{c:34s} {'OK' if r.returncode == 0 else 'CRASH rc=%d' % r.returncode}  {r.stderr.strip().splitlines()[-1][:120] if r.returncode and r.stderr.strip() else ''}", flush=True)
