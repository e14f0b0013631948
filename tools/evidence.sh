#!/bin/bash
# The round's profile evidence in one go on the GPU box -> gpurun_out/<tag>/ (copy what is to be judged into profiles/).
#   bash tools/evidence.sh <tag>
# Passes (rocprofv3 gets `python3 <script>` directly after `--`; counters in their own runs, never with a trace):
#   1  aggregation path alone (tools/pmc_daf_frame.py): kernel trace, FETCH_SIZE pass, WRITE_SIZE pass
#        -> daf_kernels_by_grid.txt, daf_pmc_traffic.json
#   2  replayed training step (tools/try_graph_frame.py): kernel trace + MFMA counter pass -> linear_path_mfma_pmc.json
#   3  bench.py itself under --kernel-trace --stats -> bench_kernel_stats_hipad.csv + the bench line it printed
set -o pipefail
tag=${1:-evidence}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ev_*
echo "[evidence] 1 aggregation path" | tee -a "$out/evidence.log"
rocprofv3 --kernel-trace --output-format csv -d /tmp/ev_daf_trace -- python3 "$root/tools/pmc_daf_frame.py" 10 > "$out/daf_trace.log" 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/ev_daf_fetch -- python3 "$root/tools/pmc_daf_frame.py" 5 > "$out/pmc_fetch.log" 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/ev_daf_write -- python3 "$root/tools/pmc_daf_frame.py" 5 > "$out/pmc_write.log" 2>&1 &&
python3 "$root/tools/trace_summary.py" /tmp/ev_daf_trace 40 > "$out/daf_kernels_by_grid.txt" 2>> "$out/evidence.log" &&
python3 "$root/tools/pmc_traffic.py" /tmp/ev_daf_fetch /tmp/ev_daf_write "$out/daf_pmc_traffic.json" >> "$out/evidence.log" 2>&1
echo "[evidence] 2 MFMA counters of the replayed step" | tee -a "$out/evidence.log"
NSTEPS=6 rocprofv3 --kernel-trace --output-format csv -d /tmp/ev_step_trace -- python3 "$root/tools/try_graph_frame.py" > "$out/step_trace.log" 2>&1 &&
NSTEPS=6 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/ev_step_pmc -- python3 "$root/tools/try_graph_frame.py" > "$out/step_pmc.log" 2>&1 &&
python3 "$root/tools/pmc_mfma.py" /tmp/ev_step_pmc /tmp/ev_step_trace "$out/linear_path_mfma_pmc.json" > "$out/pmc_mfma.log" 2>&1
echo "[evidence] 3 bench.py under --kernel-trace --stats" | tee -a "$out/evidence.log"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ev_bench -- python3 "$root/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > "$out/bench_under_rocprofv3.json" 2> "$out/bench_under_rocprofv3.err"
f=$(ls /tmp/ev_bench/*/*kernel_stats.csv 2>/dev/null | head -1)
if [ -n "$f" ]; then
  head -1 "$f" > "$out/bench_kernel_stats_hipad.csv"
  grep "hipad::" "$f" >> "$out/bench_kernel_stats_hipad.csv"
  python3 "$root/tools/stats_top.py" "$f" 60 > "$out/bench_kernel_stats_top60.txt" 2>> "$out/evidence.log"
fi
tail -3 "$out/evidence.log"
