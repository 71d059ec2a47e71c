#!/bin/bash
# A/B of the four-workgroups-per-row tail kernel's variants (CUADMM_TS_VARIANT, a developer switch that exists only while a variant is
# being measured) on the inputs whose tails are beyond 18 432 columns -> gpurun_out/ts_variants.log
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
O=gpurun_out/ts_variants.log
: > $O
for V in 0 1; do
  CUADMM_TS_VARIANT=$V timeout 600 python tools/run_large.py PushT_N=30_MOMENT 11000 400 2>&1 | grep RESULT | sed "s/^/[variant $V] /" >> $O
  CUADMM_TS_VARIANT=$V timeout 600 python tools/run_large.py PushBox_N=50_MOMENT 11000 400 2>&1 | grep RESULT | sed "s/^/[N50 variant $V] /" >> $O
  CUADMM_TS_VARIANT=$V CUADMM_TAIL_K=24576 timeout 600 python tools/run_large.py PushBox_N=30_MOMENT 11000 400 2>&1 | grep RESULT | sed "s/^/[N30 k=24576 variant $V] /" >> $O
done
cat $O
