#!/bin/bash
# round 5: the small-tree LDS bound chosen at build (lead_small_kb = 0) against 16
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_lead_small_kb_auto.log
: > $O
run() { timeout 900 python tools/run_large.py "$@" 2>&1 | grep "RESULT\|resident trees" | sed -e 's/errRp.*dobj [-0-9.e+]* |//' | cut -c1-400 | tee -a $O; }
for kb in 0 16; do
  run PushBox_N=30_MOMENT 11000 1500 lead_small_kb=$kb lead_debug=1
  run PushBox_N=50_MOMENT 11000 1500 lead_small_kb=$kb lead_debug=1
  run PlanarHand_N=1_MOMENT 0 1500 lead_small_kb=$kb lead_debug=1
  run pendulum_N=80 11000 1500 lead_small_kb=$kb lead_debug=1
  run PushT_N=30_MOMENT 11000 1500 lead_small_kb=$kb lead_debug=1
  run PlanarHand_N=10_MOMENT 11000 300 lead_small_kb=$kb lead_debug=1
done
timeout 1500 python -m pytest tests/test_gpu_moment_parity.py -x -q -k "bit_for_bit or dense_tree_tops or falls_back" 2>&1 | tail -3
