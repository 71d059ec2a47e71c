#!/bin/bash
# round 5: LDS bound of the small-tree class of the leading sweeps (occupancy of the merged launch)
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_lead_small_kb.log
: > $O
run() { timeout 900 python tools/run_large.py "$@" 2>&1 | grep "RESULT\|resident trees" | sed -e 's/errRp.*dobj [-0-9.e+]* |//' | cut -c1-500 | tee -a $O; }
for kb in 16 8 4; do run PushBox_N=30_MOMENT 11000 1500 lead_small_kb=$kb lead_debug=1; done
for kb in 16 8 4; do run PushBox_N=50_MOMENT 11000 1500 lead_small_kb=$kb lead_debug=1; done
for kb in 16 8 4; do run PlanarHand_N=1_MOMENT 0 1500 lead_small_kb=$kb lead_debug=1; done
for kb in 16 8 4; do run pendulum_N=80 11000 1500 lead_small_kb=$kb lead_debug=1; done
for kb in 16 8 4; do run PlanarHand_N=10_MOMENT 11000 300 lead_small_kb=$kb lead_debug=1; done
