set -x
cd $GRAFT_REPO_ROOT
for t in 1 4 8 16; do
echo "HOST THREADS $t"
CUADMM_HOST_THREADS=$t timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['breakdown_ms_per_iter'])"
done
timeout 600 python -m pytest tests/test_gpu_solver.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3
