#!/bin/bash
# hybrid y-solve (L21 on the device, L11 sweeps on the host) against the host-only leading part -> gpurun_out/hybrid_ab.log
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/hybrid_ab.log
: > $O
python -m pytest tests/test_gpu_moment_parity.py -q -m gpu -x -k "hybrid" 2>&1 | tail -5 | tee -a $O
timeout 900 python tools/run_large.py PlanarHand_N=10_MOMENT 11000 1500 2>&1 | grep RESULT | sed "s/^/[hybrid] /" >> $O
timeout 900 python tools/run_large.py PlanarHand_N=10_MOMENT 11000 1500 l21_device=0 2>&1 | grep RESULT | sed "s/^/[host] /" >> $O
cat $O
