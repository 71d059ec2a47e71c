cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/full_pytest.log 2>&1; echo rc=$?; grep -a "passed\|failed" gpurun_out/full_pytest.log | tail -3
pl() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['breakdown_ms_per_iter'].items()}, d['roofline']['frac'])"; }
timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep '^{' | pl "c2"
timeout 300 python bench.py --config c4 --no-cpu-baseline 2>&1 | grep '^{' | pl "c4"
