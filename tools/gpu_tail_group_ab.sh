#!/bin/bash
# the tail beyond 18 432 columns: four workgroups per row (ts_onepass_group_kernel, default) against the two triangular GEMVs
# (option tail_one_pass = 0) on the inputs with such tails -> gpurun_out/tail_group_ab.log.  (The kernel's variants -- rows per exchange,
# members per row, register prefetch, occupancy -- were measured with a temporary switch in TailSolve::apply; DESIGN.md section 4 has
# the numbers.)
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
O=gpurun_out/tail_group_ab.log
: > $O
for OP in 1 0; do
  timeout 600 python tools/run_large.py PushT_N=30_MOMENT 11000 400 tail_one_pass=$OP 2>&1 | grep RESULT | sed "s/^/[one_pass=$OP] /" >> $O
  timeout 600 python tools/run_large.py PushBox_N=50_MOMENT 11000 400 tail_one_pass=$OP 2>&1 | grep RESULT | sed "s/^/[one_pass=$OP] /" >> $O
  CUADMM_TAIL_K=24576 timeout 600 python tools/run_large.py PushBox_N=30_MOMENT 11000 400 tail_one_pass=$OP 2>&1 | grep RESULT | sed "s/^/[k=24576 one_pass=$OP] /" >> $O
done
cat $O
