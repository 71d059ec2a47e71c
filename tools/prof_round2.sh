# Round-2 profiling on the GPU box: per config kernel trace + stats, then PMC passes on their own (no trace domains with --pmc)
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
G=$R/gpurun_out
prof () {   # tag, command...
  tag=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $G/prof_${tag} -- "$@" > $G/${tag}_trace.log 2>&1
  timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $G/prof_${tag}_fetch -- "$@" > $G/${tag}_fetch.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $G/prof_${tag}_write -- "$@" > $G/${tag}_write.log 2>&1
  timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $G/prof_${tag}_sq -- "$@" > $G/${tag}_sq.log 2>&1
  python3 $R/tools/summarize_prof.py r02_${tag} $(dirname $(find $G/prof_${tag} -name "*kernel_stats.csv" | head -1)) \
      $(dirname $(find $G/prof_${tag}_fetch -name "*counter_collection.csv" | head -1)) \
      $(dirname $(find $G/prof_${tag}_write -name "*counter_collection.csv" | head -1)) \
      $(dirname $(find $G/prof_${tag}_sq -name "*counter_collection.csv" | head -1)) > $G/${tag}_summary.log 2>&1
}
prof c2 python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-breakdown
prof c3 python3 $R/bench.py --config c3 --steps 10 --warmup 2 --no-cpu-baseline --no-breakdown
prof c4 python3 $R/bench.py --config c4 --steps 12 --warmup 2 --no-cpu-baseline --no-breakdown
prof planarhand python3 $R/tools/run_real.py PlanarHand_N=1_MOMENT 60 0
cd $R
cp profiles/r02_*_kernel_stats.csv profiles/r02_*_pmc_*.json gpurun_out/ 2>/dev/null
ls profiles | grep r02
# the bench lines of the round, outside the profiler
timeout 600 python3 bench.py 2>/dev/null | grep '^{' > gpurun_out/r02_bench_c2.json
timeout 600 python3 bench.py --config c3 2>/dev/null | grep '^{' > gpurun_out/r02_bench_c3.json
timeout 600 python3 bench.py --config c4 2>/dev/null | grep '^{' > gpurun_out/r02_bench_c4.json
timeout 600 python3 bench.py --mode sgs --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r02_bench_c2_sgs.json
CUADMM_BENCH_FORCE_DIST=1 timeout 600 python3 bench.py --sharding allreduce --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/r02_bench_c2_allreduce_forced_1rank.json
head -c 600 gpurun_out/r02_bench_c2.json
