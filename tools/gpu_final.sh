cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_pytest_full.log 2>&1; echo rc=$? > gpurun_out/r02_gpu_tests.log; grep -a "passed\|failed" gpurun_out/r02_gpu_pytest_full.log | tail -2 >> gpurun_out/r02_gpu_tests.log
cat gpurun_out/r02_gpu_tests.log
bash tools/prof_round2.sh > gpurun_out/r02_prof.log 2>&1
tail -5 gpurun_out/r02_prof.log
bash tools/gpu_bench_lines.sh 2>&1 | tail -8
timeout 900 python tools/run_all_real.py > gpurun_out/r02_real_data_short.log 2>&1; grep -a "|" gpurun_out/r02_real_data_short.log | cut -c1-330
