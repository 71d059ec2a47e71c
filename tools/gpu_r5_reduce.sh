#!/bin/bash
# round 5: the tail's partial-sum reduction on eight slices, the tail copy inside tops_u_kernel: tests that touch them, bench lines
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 3000 python -m pytest tests/test_gpu_ops.py tests/test_gpu_moment_parity.py tests/test_gpu_sharded_procs.py tests/test_gpu_contention.py -q -x 2>&1 | tail -4
for c in c1 c5; do
  timeout 400 python bench.py --config $c --no-cpu-baseline --no-breakdown 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c', round(d['value'],1), 'ms/step', round(d['ms_per_step'],4), 'steady', d.get('steady_state',{}).get('value'))"
done
for p in "PushBox_N=30_MOMENT 11000 1500" "PushBox_N=50_MOMENT 11000 1500" "PlanarHand_N=10_MOMENT 11000 300"; do timeout 600 python tools/run_large.py $p 2>&1 | grep RESULT | grep -o "RESULT [A-Za-z_=0-9]*\|[0-9.]* ms/iter\|'tail_solve': np.float64([0-9.]*)" | paste - - -; done
