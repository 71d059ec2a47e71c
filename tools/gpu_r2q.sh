cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_solver.py -x -q 2>&1 | grep "passed\|failed\|Error" | tail -3
timeout 900 python tools/run_all_real.py 2>&1 | grep "PlanarHand\|pendulum\|PushT" | cut -c1-360
