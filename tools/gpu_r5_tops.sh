#!/bin/bash
# round 5: dense tree tops (lead_solve.h) -- the small agreement tests, then PlanarHand_N=10 with the cut forest against the hybrid solve
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_tops.log
: > $O
timeout 900 python -m pytest tests/test_gpu_moment_parity.py -x -q -k "dense_tree_tops" 2>&1 | tail -15 | tee -a $O
for opt in "lead_debug=1" "lead_tops=0"; do
  timeout 900 python tools/run_large.py PlanarHand_N=10_MOMENT 11000 300 $opt 2>&1 | grep "RESULT\|lead debug\|y-solve" | cut -c1-900 | tee -a $O
done
