"""MFMA utilisation of the linear path from one `rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES
SQ_BUSY_CYCLES GRBM_GUI_ACTIVE` pass of the bench command plus a kernel trace of the same command (durations are taken
from the un-instrumented run: a profiled pass runs at a lower clock).

    python tools/pmc_mfma.py <pmc_dir> <trace_dir> <out.json>

Counter units (MI355X_MICROARCH.md, "Per-instruction cycle constants" / "DVFS give-back"):
  SQ_INSTS_VALU_MFMA_MOPS_BF16   512 flops per count, summed over the chip
  SQ_VALU_MFMA_BUSY_CYCLES       shader cycles a SIMD's matrix pipe is busy, summed over the chip's 1024 SIMDs
                                 (16 per v_mfma_f32_16x16x32_bf16, 32 per 32x32x16: `busy_cycles_per_16k_flop` checks it)
  GRBM_GUI_ACTIVE                rocprofv3 reports the SUM over the 8 XCDs: the launch's active cycles are GUI / 8
so   mfma_util = BUSY / (GUI / 8 x 1024 SIMDs)   and   effective clock = GUI / 8 / duration.
The two views of one launch agree when  frac_of_2p5PF ~= mfma_util x effective_clock / 2.4 GHz  (2.5 PFLOP/s is 1024
SIMDs x 16384 flop / 16 cycles at 2.4 GHz); `self_consistency` is the ratio of the two sides (1.0 = agree; the round-2
table divided by GUI instead of GUI / 8 and read 8x low).
"""
import collections, csv, glob, json, sys

SIMDS = 256 * 4
PEAK_TFLOPS = 2500.0
KERNELS = ("hipad::gemm_kernel", "hipad::gemm_fwd_hilo_kernel", "hipad::linear_bwd_fused_kernel", "hipad::chain_fwd_kernel", "hipad::chain_bwd_kernel",
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
    mfma16k = flops / 16384.0                       # MFMA instructions, in units of a 16x16x32 bf16 one
    util = busy / (gui / 8.0 * SIMDS) if gui else None
    clock = gui / 8.0 / (us * 1e-6) / 1e9 if (gui and us) else None
    frac = None if not us else flops / (us * 1e-6) / 1e12 / PEAK_TFLOPS
    out[k] = dict(launches_profiled=n, mfma_gflop_per_launch=round(flops / 1e9, 4),
                  busy_cycles_per_16k_flop=round(busy / mfma16k, 2) if mfma16k else None,
                  mfma_util_pct=round(100.0 * util, 3) if util is not None else None,
                  effective_clock_GHz=None if clock is None else round(clock, 3),
                  avg_us=None if us is None else round(us, 2),
                  achieved_tflops=None if not us else round(flops / (us * 1e-6) / 1e12, 2),
                  frac_of_2p5PF=None if frac is None else round(frac, 5),
                  self_consistency=None if not (util and clock and frac) else round(frac / (util * clock / 2.4), 3))
json.dump(dict(note="per launch averages; MFMA flops = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512; util = SQ_VALU_MFMA_BUSY_CYCLES / "
                    "(GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); effective clock = GUI / 8 / duration; durations from the "
                    "un-instrumented kernel trace of the same command; self_consistency = frac_of_2p5PF / (util x clock / 2.4 GHz)",
               kernels=out), open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
