set -x
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py 2>&1 | grep '^{' > gpurun_out/r2e_bench_c2.json; cut -c1-3000 gpurun_out/r2e_bench_c2.json
timeout 600 python bench.py --config c4 2>&1 | grep '^{' > gpurun_out/r2e_bench_c4.json; cut -c1-3000 gpurun_out/r2e_bench_c4.json
CUADMM_BENCH_FORCE_DIST=1 timeout 600 python bench.py --sharding allreduce --no-cpu-baseline 2>&1 | grep '^{' | cut -c1-2500
CUADMM_BENCH_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline 2>&1 | grep '^{' | cut -c1-600
timeout 600 python bench.py --mode sgs --no-cpu-baseline 2>&1 | grep '^{' | cut -c1-600
