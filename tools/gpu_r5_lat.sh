#!/bin/bash
# round 5, latency items: copy kernel, merged lead sweeps, spin-wait -- tests of the paths + A/B on c1 / c5 / c2
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_moment_parity.py tests/test_gpu_solver.py tests/test_gpu_mode_matrix.py tests/test_gpu_sharded.py -q -x > gpurun_out/r05_lat_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r05_lat_tests.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/r05_lat_tests.log | tail -6
for opt in "" "--option spin_wait=0"; do
for c in c1 c5 c2 c3; do
  timeout 400 python bench.py --config $c --no-cpu-baseline --no-breakdown $opt > gpurun_out/ab.json 2>/dev/null
  python -c "
import json
d=json.load(open('gpurun_out/ab.json'))
print('$c', '$opt', round(d['value'],1), 'steady', round(d.get('steady_state',{}).get('value',0),1))
"
done
done
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/ab.json')); print('driver', d['value'])"; done
