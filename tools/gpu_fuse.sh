cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_solver.py tests/test_gpu_configs.py -x -q 2>&1 | grep -a "passed\|failed\|Error\|assert" | tail -8
pl() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['breakdown_ms_per_iter'].items()}, d['roofline']['frac'])"; }
timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep '^{' | pl "c2 fused gen4"
CUADMM_PSD_W32_GEN=3 timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep '^{' | pl "c2 fused gen3"
CUADMM_FUSE=0 timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep '^{' | pl "c2 unfused"
timeout 300 python bench.py --mode sgs --no-cpu-baseline 2>&1 | grep '^{' | pl "c2 sgs fused"
timeout 300 python bench.py --config c4 --no-cpu-baseline 2>&1 | grep '^{' | pl "c4 fused"
CUADMM_FUSE=0 timeout 300 python bench.py --config c4 --no-cpu-baseline 2>&1 | grep '^{' | pl "c4 unfused"
