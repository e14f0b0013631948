"""The aggregation path of a stage-2 frame as the training step runs it (24 forwards, 24 grad loc+weights launches, ONE
merged feature-gradient pass), a few frames, nothing else -- the program to put behind `rocprofv3 --pmc FETCH_SIZE` /
`--pmc WRITE_SIZE` / `--kernel-trace` (tools/pmc_traffic.py, tools/trace_summary.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import hipad_amd  # noqa
import torch
import bench

daf = bench.DafStage2(torch.device("cuda", 0), seed=0)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    daf.step()
torch.cuda.synchronize()
print("frame feature-gradient pass: algorithmic bytes", daf.frame_feat_alg_bytes(), "rows", daf.rows_touched_frame)
for d in daf.calls:
    print(d["name"], "fwd", daf.alg_bytes(d, "fwd"), "bwd_lw", daf.alg_bytes(d, "bwd_lw"), "kept pairs", d["kept_pairs"], "rows", d["rows_touched"])
