#!/bin/bash
# round 5: c1 with the schedule's warm start on the batched-GEMM path too (psd_hint = 2)
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for o in "psd_hint=1" "psd_hint=2" "psd_hint=0"; do
  timeout 600 python bench.py --config c1 --steps 300 --warmup 20 --no-cpu-baseline --no-breakdown --time-to-tol 0 --option $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$o', round(d['value'],1), d['roofline']['newton_schulz_steps'], round(d['roofline']['avg_launch_ms'],4))"
done
