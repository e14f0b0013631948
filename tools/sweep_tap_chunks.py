"""Frame feature-gradient pass (hipad_daf_backward_feat_multi over the 24 calls of a stage-2 frame) against the work
split of its two tap passes (hipad_daf_set_tap_chunks).  GPU box.

    python tools/sweep_tap_chunks.py [chunks ...] [run=R ...]

``run=R``: sweep hipad_daf_set_feat_run (batches of 64 sorted taps per wave run of the accumulation pass) instead.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import hipad_amd  # noqa
import torch
import bench

daf = bench.DafStage2(torch.device("cuda", 0), seed=0)
lib = daf.lib.load()

def timed():
    daf.bwd_feat_frame()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        daf.bwd_feat_frame()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


blocks = [int(v[7:]) for v in sys.argv[1:] if v.startswith("blocks=")]
for b in blocks:
    lib.hipad_daf_set_feat_blocks(b)
    print("blocks %5d: %.1f us per frame pass" % (b, timed()), flush=True)
if blocks:
    lib.hipad_daf_set_feat_blocks(0)
    sys.exit(0)
runs = [int(v[4:]) for v in sys.argv[1:] if v.startswith("run=")]
for r in runs:
    lib.hipad_daf_set_feat_run(r)
    print("run %3d batches: %.1f us per frame pass" % (r, timed()), flush=True)
if runs:
    lib.hipad_daf_set_feat_run(0)
    sys.exit(0)
for nch in [int(v) for v in sys.argv[1:]] or [0, 2, 4, 8, 16, 24, 32, 48, 64]:
    lib.hipad_daf_set_tap_chunks(nch)
    daf.bwd_feat_frame()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        daf.bwd_feat_frame()
    e1.record(); e1.synchronize()
    print("chunks %3d (0 = auto): %.1f us per frame pass" % (nch, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
lib.hipad_daf_set_tap_chunks(0)
