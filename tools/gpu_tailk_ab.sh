#!/bin/bash
# the tail planner's deep-forest branch (aat_ldlt.cpp plan_tail) against forced tail sizes on the PushBox inputs -> gpurun_out/tailk_ab.log
# usage: bash tools/gpu_tailk_ab.sh [parity]
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/tailk_ab.log
: > $O
run() { K=$1; shift; CUADMM_TAIL_K=$K timeout 900 python tools/run_large.py "$@" lead_debug=1 2>&1 | grep -E "RESULT|lead debug" | sed "s/^/[k=$K] /" >> $O; }
run -1 PushBox_N=30_MOMENT 11000 60000
run -1 PushBox_N=30_MOMENT 0 60000
run -1 PushBox_N=50_MOMENT 11000 60000
run 10240 PushBox_N=30_MOMENT 11000 60000
cat $O
if [ "$1" = parity ]; then
  python -m pytest tests/test_gpu_moment_parity.py tests/test_gpu_longrun.py -q -m gpu -x 2>&1 | tail -15 | tee -a $O
fi
