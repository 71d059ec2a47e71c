cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_psd.py tests/test_gpu_fused.py -x -q > gpurun_out/step_pytest.log 2>&1; grep -a "passed\|failed" gpurun_out/step_pytest.log | tail -2
pl() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v,3) for k,v in d['breakdown_ms_per_iter'].items()}, round(d['roofline']['frac'],4))"; }
timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep '^{' | pl "c2"
timeout 300 python bench.py --config c4 --no-cpu-baseline 2>&1 | grep '^{' | pl "c4"
timeout 300 python bench.py --mode sgs --no-cpu-baseline 2>&1 | grep '^{' | pl "c2 sgs"
timeout 600 python tools/run_all_real.py 2>&1 | grep "PlanarHand\|pendulum\|PushT\|ros_2000" | cut -c1-330
