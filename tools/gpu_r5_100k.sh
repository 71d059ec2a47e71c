#!/bin/bash
# round 5: pendulum N = 80, the whole 100 000 iterations of the reference's log against the oracle's own 100 000th row
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_moment_parity.py -q -s -k "100000" 2>&1 | grep -v "^$" | cut -c1-700 | tail -12 | tee gpurun_out/r05_pendulum_100k.log
