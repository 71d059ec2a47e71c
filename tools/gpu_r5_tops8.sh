#!/bin/bash
# round 5: dense tree tops -- height of the cut (the planner's tail, 1 500 iterations each so that the clock has settled)
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_tops_level.log
: > $O
run() { timeout 900 python tools/run_large.py "$@" 2>&1 | grep "RESULT" | sed -e 's/errRp.*dobj [-0-9.e+]* |//' | cut -c1-500 | tee -a $O; }
for L in 12 16 24 32 48; do run PushBox_N=30_MOMENT 11000 1500 tail_k=8448 lead_tops=$L; done
for L in 16 24 32 48; do run PushBox_N=50_MOMENT 11000 1500 tail_k=8448 lead_tops=$L; done
for L in 16 24 32 48; do run PlanarHand_N=1_MOMENT 0 1500 tail_k=10752 lead_tops=$L; done
for L in 16 24 32; do run PushT_N=30_MOMENT 11000 1500 tail_k=16384 lead_tops=$L; done
for L in 24 48; do run PlanarHand_N=10_MOMENT 11000 300 lead_tops=$L; done
