#!/bin/bash
# round 5 experiment: PlanarHand_N=10 (m = 483 707) with the GPU tail beyond its 32 768-column cap (tail_max_k / forced tail_k)
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
free -g | head -2; cat /sys/fs/cgroup/memory.max 2>/dev/null
for opt in "" "tail_k=40960" "tail_k=49152"; do
  timeout 900 python tools/run_large.py PlanarHand_N=10_MOMENT 11000 300 $opt 2>&1 | grep RESULT | cut -c1-700
done
