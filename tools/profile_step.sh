#!/bin/bash
# Kernel trace of the replayed training step on the GPU box -> per-category / per-kernel / per-grid summaries.
#   bash tools/profile_step.sh <tag>      writes gpurun_out/<tag>/{categories,launch_counts,kernels_by_grid}.txt
# (rocprofv3 gets the program itself after `--`: python3 <script>, no wrapper.)
set -o pipefail
tag=${1:-prof}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
NSTEPS=${NSTEPS:-8} rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -- python3 "$root/tools/try_graph_frame.py" > "$out/graph_frame_traced.log" 2>&1
cd "$root"
python tools/trace_categories.py /tmp/prof_$tag steps 3 > "$out/categories.txt"
python tools/trace_counts.py /tmp/prof_$tag steps 3 90 > "$out/launch_counts.txt"
python tools/trace_summary.py /tmp/prof_$tag 160 > "$out/kernels_by_grid.txt" 2> "$out/kernels_by_grid.err"
grep -v "^W2\|^E2" "$out/graph_frame_traced.log" | tail -4
cat "$out/categories.txt"
