#!/bin/bash
# The pinned replay test (tests/test_graph_step_gpu.py::test_replayed_step_tracks_eager_step) after other test files in
# ONE process: prints its eager-vs-eager noise floor per combination ("Open observation [r3]").
#   bash tools/bisect_test_order.sh            # the known-bad order first, then halves of it
# GPU box; every combination is its own pytest process (one at a time).
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$root" && mkdir -p gpurun_out/bisect
P=tests/test_graph_step_gpu.py::test_replayed_step_tracks_eager_step
run() {
  name=$1; shift
  python -m pytest "$@" "$P" -q -m gpu -s -p no:cacheprovider > gpurun_out/bisect/$name.txt 2>&1
  echo "== $name: $(grep -c passed gpurun_out/bisect/$name.txt) summary line(s)"
  grep "eager vs eager\|passed\|failed" gpurun_out/bisect/$name.txt | cut -c1-220
}
run bad_order tests/test_depth_loss_gpu.py tests/test_losses.py tests/test_encoder.py
run depth_losses tests/test_depth_loss_gpu.py tests/test_losses.py
run losses_encoder tests/test_losses.py tests/test_encoder.py
run depth_encoder tests/test_depth_loss_gpu.py tests/test_encoder.py
run losses_all tests/test_losses.py
