"""MFMA utilisation of the linear path from one `rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES
SQ_BUSY_CYCLES GRBM_GUI_ACTIVE` pass of the bench command: per hand-written MFMA kernel the launches, bf16 MFMA flops
(MOPS x 512), busy fraction of the matrix pipes (MFMA busy cycles / (GRBM_GUI_ACTIVE x 4 SIMDs x 256 CUs / 8 XCDs ...
see below) and, with a kernel-trace of the same command, the achieved TFLOP/s against the 2.5 PFLOP/s dense bf16 peak.

    python tools/pmc_mfma.py <pmc_dir> <trace_dir> <out.json>

MfmaUtil as rocprofv3 defines it for gfx94x/gfx950: SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip) /
(GRBM_GUI_ACTIVE (max over XCDs) x SIMD_NUM), SIMD_NUM = 256 CUs x 4.
"""
import collections, csv, glob, json, sys

SIMDS = 256 * 4
PEAK_TFLOPS = 2500.0
KERNELS = ("hipad::gemm_kernel", "hipad::linear_bwd_fused_kernel", "hipad::chain_fwd_kernel", "hipad::chain_bwd_kernel",
           "hipad::chain_dw_kernel", "hipad::attn_fwd_kernel", "hipad::attn_bwd_dq_kernel", "hipad::attn_bwd_dkv_kernel")


def short(name):
    name = name.replace("void ", "")
    for k in KERNELS:
        if name.startswith(k):
            return name.split("(")[0][:70]
    return None


pmc = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
with open(pmc) as fh:
    for r in csv.DictReader(fh):
        k = short(r["Kernel_Name"])
        if k:
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
trace = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
with open(trace) as fh:
    for r in csv.DictReader(fh):
        k = short(r["Kernel_Name"])
        if k:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {}
for k, c in sorted(vals.items()):
    n = len(c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", [])) or 1
    flops = 512.0 * sum(c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", [0])) / n
    busy = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) / n
    gui = sum(c.get("GRBM_GUI_ACTIVE", [0])) / n
    us = sum(dur[k]) / len(dur[k]) if dur.get(k) else None
    out[k] = dict(launches_profiled=n, mfma_gflop_per_launch=round(flops / 1e9, 4),
                  mfma_util_pct=round(100.0 * busy / (gui * SIMDS), 3) if gui else None,
                  avg_us=None if us is None else round(us, 2),
                  achieved_tflops=None if not us else round(flops / (us * 1e-6) / 1e12, 2),
                  frac_of_2p5PF=None if not us else round(flops / (us * 1e-6) / 1e12 / PEAK_TFLOPS, 5))
json.dump(dict(note="per launch averages; MFMA flops = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512; util = SQ_VALU_MFMA_BUSY_CYCLES / "
                    "(GRBM_GUI_ACTIVE x 1024 SIMDs); durations from the un-instrumented kernel trace of the same command",
               kernels=out), open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
