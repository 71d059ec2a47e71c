#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
O=gpurun_out/ts_variants.log
: > $O
for V in 1 2 3; do
  CUADMM_TS_VARIANT=$V timeout 600 python tools/run_large.py PushT_N=30_MOMENT 11000 400 2>&1 | grep RESULT | sed "s/^/[variant $V] /" >> $O
done
for V in 1 2; do
  CUADMM_TS_VARIANT=$V timeout 600 python tools/run_large.py PushBox_N=50_MOMENT 11000 400 2>&1 | grep RESULT | sed "s/^/[N50 variant $V] /" >> $O
done
cat $O
