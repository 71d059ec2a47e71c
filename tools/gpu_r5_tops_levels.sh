#!/bin/bash
# round 5: dense tree tops at many cut heights and tail sizes against the plain solve at the same tail (tools/probe_tops_levels.py)
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 3000 python tools/probe_tops_levels.py 2>&1 | tee gpurun_out/r05_tops_levels_sweep.log | tail -60
