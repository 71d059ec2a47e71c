#!/bin/bash
# round 5: the reference's largest shipped problems to the reference CLI's tolerance (1e-3), init broken out -> gpurun_out/r05_real_data.log
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_real_data.log
: > $O
timeout 600 python tools/run_large.py PushBox_N=30_MOMENT 11000 60000 2>&1 | grep RESULT >> $O
timeout 900 python tools/run_large.py PushBox_N=30_MOMENT 0 60000 2>&1 | grep RESULT >> $O
timeout 900 python tools/run_large.py PushBox_N=50_MOMENT 11000 60000 2>&1 | grep RESULT >> $O
timeout 1500 python tools/run_large.py PlanarHand_N=10_MOMENT 11000 25000 2>&1 | grep RESULT >> $O
timeout 600 python tools/run_large.py PushT_N=30_MOMENT 11000 60000 2>&1 | grep RESULT >> $O
timeout 600 python tools/run_large.py PushBox_N=30_MOMENT 11000 60000 lead_tops=0 2>&1 | grep RESULT >> $O
cat $O | cut -c1-420
bash tools/prof_round5.sh c1 2>&1 | tail -5
