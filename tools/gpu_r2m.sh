cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_psd.py tests/test_gpu_configs.py tests/test_gpu_solver.py -x -q 2>&1 | tail -3
for c in c2 c3 c4; do timeout 300 python bench.py --config $c --no-cpu-baseline --no-breakdown 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],4), d['roofline']['newton_schulz_steps']['mean'])"; done
