#!/bin/bash
# round 5 experiment: pivots of the GPU tail below a threshold treated as zero (tail_pinv_tol) -- deviation from the oracle trajectories
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_pinv.log
: > $O
timeout 1200 python tools/probe_moment.py "PushT_N=30_MOMENT/switch=11000" ";tail_pinv_tol=1e-12;tail_pinv_tol=1e-10" 2>&1 | cut -c1-420 | tee -a $O
timeout 1200 python tools/probe_moment.py "PushBox_N=50_MOMENT/switch=11000" "tail_k=8448,lead_tops=32;tail_k=8448,lead_tops=32,tail_pinv_tol=1e-12" 2>&1 | cut -c1-420 | tee -a $O
timeout 1200 python tools/probe_moment.py "PlanarHand_N=10_MOMENT/switch=11000" ";tail_pinv_tol=1e-12" 2>&1 | cut -c1-420 | tee -a $O
