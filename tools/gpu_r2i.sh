cd $GRAFT_REPO_ROOT
timeout 2400 python -X faulthandler -m pytest tests/test_gpu_longrun.py -q -s > gpurun_out/r2i_pytest.log 2>&1
echo "pytest rc $?"
grep "longrun\|passed\|failed\|Error\|assert" gpurun_out/r2i_pytest.log | cut -c1-400 | head -60
