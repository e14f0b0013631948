"""Are hipMemsetAsync nodes honoured on every replay of a captured graph?  (GPU box)"""
import torch
dev = "cuda"
for n in (1 << 10, 1 << 16, 1 << 20, 1 << 21, 1 << 24):
    for dtype in (torch.float32, torch.bfloat16):
        buf = torch.full((n,), 7.0, device=dev, dtype=dtype)
        out = torch.zeros(n, device=dev, dtype=dtype)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            buf.zero_()             # hipMemsetAsync -> memset node
            buf.add_(1)
            tmp = torch.zeros(n, device=dev, dtype=dtype)  # pool allocation + memset node
            tmp.add_(buf)
            out.copy_(tmp)
        bad = []
        for i in range(6):
            junk = torch.full((n,), float("nan"), device=dev)
            del junk
            g.replay()
            torch.cuda.synchronize()
            if not bool((out == 1).all()) or not bool((buf == 1).all()):
                bad.append((i, float(out.float().min()), float(out.float().max()), float(buf.float().max())))
        print(n, dtype, "OK" if not bad else bad, flush=True)
