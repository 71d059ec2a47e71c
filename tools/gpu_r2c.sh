set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2c_c3 -- python3 $R/tools/run_config.py c3 2000 30 > $R/gpurun_out/r2c_c3.log 2>&1
find $R/gpurun_out/prof_r2c_c3 -name "*kernel_stats.csv" | head -1 | xargs head -20
