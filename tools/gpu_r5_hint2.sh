#!/bin/bash
# round 5: the schedule's warm start on the batched-GEMM path (psd_hint = 2) on the inputs with mid-size blocks
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_hint2.log
: > $O
run() { timeout 900 python tools/run_large.py "$@" 2>&1 | grep "RESULT" | sed -e 's/errRp.*dobj [-0-9.e+]* |//' | cut -c1-400 | tee -a $O; }
for h in 1 2; do
  run PlanarHand_N=10_MOMENT 11000 300 psd_hint=$h
  run PushBox_N=50_MOMENT 11000 1500 psd_hint=$h
  run PushT_N=30_MOMENT 11000 1500 psd_hint=$h
  run taha1a 11000 1500 psd_hint=$h
  run PlanarHand_N=1_MOMENT 0 1500 psd_hint=$h
done
