#!/bin/bash
# round 5: the small real-data examples with the reference's CLI parameters (regression check of the late changes)
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 1500 python tools/run_all_real.py 2>&1 | grep -v "^$" | cut -c1-330 | tee gpurun_out/r05_real_small.log
