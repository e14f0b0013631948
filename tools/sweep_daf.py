"""Sweep the pairs-per-wave knob of the aggregation kernels on the stage-2 call shapes (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import DafStage2
from hipad_amd import lib

wl = DafStage2(torch.device("cuda", 0), 0)
L = lib.load()
for ppw in (0, 128, 64, 32, 24, 12):
    L.hipad_daf_set_pairs_per_wave(ppw, ppw)
    kt = wl.kernel_times(reps=30)
    print(ppw, {f"{k[0]}_{k[1]}": round(v * 1e3, 1) for k, v in kt.items()}, flush=True)
