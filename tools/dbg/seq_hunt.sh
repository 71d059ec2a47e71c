cd $GRAFT_REPO_ROOT
export MALLOC_CHECK_=3
fails=0
for i in $(seq 1 14); do
  timeout 900 python -X faulthandler -m pytest tests/test_gpu_ops.py tests/test_gpu_solver.py -k "tail or test_gpu_solver or trajectory or printed or warm or rejects or converges or planarhand or duo or device_side" -q -m gpu -p no:cacheprovider > /tmp/h.log 2>&1; rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails+1)); grep -a -v "dist-packages\|^$" /tmp/h.log | head -12 | cut -c1-200; fi
done
echo "tail ops + solver with staged copies: $fails of 14 runs failed"
