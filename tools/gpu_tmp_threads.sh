#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/threads_ab.log
: > $O
nproc >> $O; lscpu | grep -E "Model name|Socket|Core|Thread|L3|NUMA node\(s\)" >> $O; free -g | head -2 >> $O
for T in 8 16; do
  CUADMM_HOST_THREADS=$T timeout 600 python tools/run_large.py PlanarHand_N=10_MOMENT 11000 400 2>&1 | grep RESULT | sed "s/^/[T=$T] /" >> $O
done
cat $O
