#!/bin/bash
# round 5: dense tree tops -- which tail size once the tops are cheap?  (calibration of plan_tail's model)
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_tops_scan.log
: > $O
run() { timeout 600 python tools/run_large.py "$@" 2>&1 | grep "RESULT\|dense tree tops" | sed -e 's/errRp.*pobj/pobj/' | cut -c1-700 | tee -a $O; }
for k in 4096 6144 8192 12288; do run PushBox_N=30_MOMENT 11000 300 tail_k=$k lead_tops=32 lead_debug=1; done
for k in 6144 8192 10240 16384; do run PushBox_N=50_MOMENT 11000 300 tail_k=$k lead_tops=32 lead_debug=1; done
run PushT_N=30_MOMENT 11000 300 lead_tops=32 lead_debug=1
for k in 8192 12288 16384 20480; do run PushT_N=30_MOMENT 11000 300 tail_k=$k lead_tops=32 lead_debug=1; done
run PlanarHand_N=1_MOMENT 0 300
for k in 6144 8192 12288; do run PlanarHand_N=1_MOMENT 0 300 tail_k=$k lead_tops=16 lead_debug=1; done
run PlanarHand_N=1_MOMENT 0 300 tail_k=8192 lead_tops=32 lead_debug=1
run pendulum_N=80 11000 300
for k in 4096 6144 8192; do run pendulum_N=80 11000 300 tail_k=$k lead_tops=16 lead_debug=1; done
for k in 16384 20480; do run PlanarHand_N=10_MOMENT 11000 100 tail_k=$k lead_tops=32 lead_debug=1; done
