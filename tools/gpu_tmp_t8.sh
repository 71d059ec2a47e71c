#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
O=gpurun_out/t8_ab.log
: > $O
for rep in 1 2; do for T in 8 16; do for c in c1 c5; do
  CUADMM_HOST_THREADS=$T timeout 600 python bench.py --config $c --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('T=$T $c', round(d['value'],1), d['ms_per_step'])" >> $O
done; done; done
cat $O
