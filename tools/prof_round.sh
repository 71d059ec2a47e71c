set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/bench.py --steps 300 --warmup 30 > $R/gpurun_out/r1d_bench.json 2> $R/gpurun_out/r1d_bench.err
tail -c 1500 $R/gpurun_out/r1d_bench.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1d -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/r1d_bench_prof.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_r1d_fetch -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r1d_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_r1d_write -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r1d_pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $R/gpurun_out/prof_r1d_sq -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r1d_pmc_sq.log 2>&1
ls $R/gpurun_out/prof_r1d* | head -30
