#!/bin/bash
# Round-4 closing evidence on the GPU box: full -m gpu suite, the bench lines of every BASELINE config, rocprofv3 summaries.
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu -s > gpurun_out/r04_gpu_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r04_gpu_tests.log
grep -E "passed|failed|rc=|^FAILED|^ERROR" gpurun_out/r04_gpu_tests.log | tail -8
line() { name=$1; shift
  timeout 900 python bench.py "$@" 2> gpurun_out/r04_bench_$name.err | grep '^{' > gpurun_out/r04_bench_$name.json
  python - gpurun_out/r04_bench_$name.json $name <<'P'
import json,sys
try:
    d=json.load(open(sys.argv[1])); r=d["roofline"]
    print("%-14s %9.1f %s  ms/step %.4f frac %.4f" % (sys.argv[2], d["value"], d["unit"][:12], d["ms_per_step"], r["frac"]), d.get("breakdown_ms_per_iter"), (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[2], "ERR", e)
P
}
line c2 --steps 200 --warmup 20
line c2_driver --steps 20 --warmup 5
line c2_sgs --mode sgs --no-cpu-baseline
line c2_sparse --c-sparse --no-cpu-baseline
line c2_proj --projection-only
line c3 --config c3
line c4 --config c4
line c4_unfused --config c4 --option fuse=0 --no-cpu-baseline
line c4_proj --config c4 --projection-only --steps 20 --warmup 3
line c1 --config c1
line c5 --config c5
CUADMM_BENCH_FORCE_DIST=1 line c2_allreduce_forced_1rank --sharding allreduce --no-cpu-baseline
timeout 300 python tools/probe_rampup.py > gpurun_out/r04_clock_ramp.log 2>&1; tail -2 gpurun_out/r04_clock_ramp.log | cut -c1-300
# two ranks on this one GPU through torch.distributed.run (gloo transport): the line that carries both exchange patterns
CUADMM_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29688 bench.py --gpus 2 --no-cpu-baseline --blocks-per-gpu 5000 --steps 40 --warmup 10 2>gpurun_out/r04_bench_c2_two_ranks.err | grep '^{' > gpurun_out/r04_bench_c2_two_ranks_one_gpu.json
python -c "import json; d=json.load(open('gpurun_out/r04_bench_c2_two_ranks_one_gpu.json')); print('two ranks on one GPU: value', d['value'], 'allreduce_path', d['allreduce_path']['value'], d['allreduce_path']['allreduce_ms_per_iter'])"
for u in mfma_coissue mem_trip gemm_sym48; do (cd /tmp && hipcc --offload-arch=gfx950 -O3 $GRAFT_REPO_ROOT/tools/ubench/$u.hip -o /tmp/$u.exe 2>/dev/null && timeout 200 /tmp/$u.exe) > gpurun_out/r04_ubench_$u.log 2>&1; done
bash tools/prof_round4.sh c2 c2_20 c3 c4 c1 c5 2>&1 | tail -30
