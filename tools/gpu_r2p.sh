cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_configs.py tests/test_gpu_sharded.py tests/test_gpu_sharded_procs.py tests/test_f4_free_and_rank.py tests/test_gpu_mex.py -x -q 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['breakdown_ms_per_iter'].items()})"; done
CUADMM_NO_FUSED_STATS=1 timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nofuse', round(d['value'],1), round(d['ms_per_step'],4))"
