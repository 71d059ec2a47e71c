#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_moment_parity.py tests/test_gpu_sharded_procs.py tests/test_gpu_mode_matrix.py tests/test_hpp_facade.py tests/test_gpu_batch.py tests/test_gpu_fused.py tests/test_gpu_solver.py tests/test_gpu_configs.py -q -m gpu -x > gpurun_out/r3m_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r3m_tests.log
grep -E "passed|failed|rc=|^FAILED|^ERROR|^E  " gpurun_out/r3m_tests.log | cut -c1-300 | tail -20
timeout 300 python bench.py --config c4 --batch 0 --option fuse=0 --no-cpu-baseline > gpurun_out/r3m_c4_unfused.json 2> gpurun_out/r3m_c4_unfused.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r3m_c4_unfused.json"))
print("c4 unfused", d["value"], d["breakdown_ms_per_iter"], "L", d["config"]["vec_len"])
P
