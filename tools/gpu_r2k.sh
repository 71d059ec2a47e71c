cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_psd.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3
timeout 300 python tools/probe_sign.py 45 3000 2>&1 | tail -5
timeout 300 python tools/probe_sign.py 64 2000 2>&1 | tail -5
timeout 300 python bench.py --config c4 --no-cpu-baseline 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['breakdown_ms_per_iter'])"
