#!/bin/bash
# round 5: early y-solve (solve_next) -- parity tests of the paths it touches, A/B on c1 / c5, the K = 33 000 tail op
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2700 python -m pytest tests/test_gpu_moment_parity.py tests/test_gpu_solver.py tests/test_gpu_mode_matrix.py tests/test_gpu_sharded.py tests/test_gpu_sharded_procs.py tests/test_gpu_longrun.py tests/test_gpu_ops.py -q -x > gpurun_out/r05_early_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r05_early_tests.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/r05_early_tests.log | tail -6
for opt in "" "--option solve_next=0"; do
for c in c1 c5; do
  timeout 400 python bench.py --config $c --no-cpu-baseline --no-breakdown $opt > gpurun_out/ab.json 2>/dev/null
  python -c "
import json
d=json.load(open('gpurun_out/ab.json'))
print('$c', '$opt', round(d['value'],1), 'steady', round(d.get('steady_state',{}).get('value',0),1))
"
done
done
timeout 400 python bench.py --config c1 --mode sgs --no-cpu-baseline --no-breakdown > gpurun_out/ab.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/ab.json')); print('c1 sgs', d['value'])"
timeout 400 python bench.py --config c1 --mode sgs --no-cpu-baseline --no-breakdown --option solve_next=0 > gpurun_out/ab.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/ab.json')); print('c1 sgs solve_next=0', d['value'])"
