cd $GRAFT_REPO_ROOT
timeout 1800 python -X faulthandler -m pytest tests -m gpu -x -q > gpurun_out/r2h_pytest.log 2>&1
echo "pytest rc $?"
grep -n "passed\|failed\|FAILED\|Fatal\|Segmentation\|core" gpurun_out/r2h_pytest.log | head
tail -c 3000 gpurun_out/r2h_pytest.log
