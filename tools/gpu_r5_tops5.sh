#!/bin/bash
# round 5: dense tree tops -- deviation from the oracle trajectory against tail size and cut height (PushBox_N=50)
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python tools/probe_moment.py "$1" "$2" 2>&1 | cut -c1-420 | tee gpurun_out/r05_tops_dev.log
