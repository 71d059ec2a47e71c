#!/bin/bash
# round 5: dense tree tops as the planner's default -- the large fixtures with the default plan, c1 / c5 bench lines, then the whole GPU suite
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
O=gpurun_out/r05_tops_default.log
: > $O
run() { timeout 900 python tools/run_large.py "$@" 2>&1 | grep "RESULT" | cut -c1-900 | tee -a $O; }
run PlanarHand_N=1_MOMENT 0 300
run PushBox_N=30_MOMENT 11000 300
run PushBox_N=50_MOMENT 11000 300
run PushT_N=30_MOMENT 11000 300
run PlanarHand_N=10_MOMENT 11000 300
for c in c1 c5; do timeout 600 python bench.py --config $c > gpurun_out/r05_tops_bench_$c.json 2> gpurun_out/r05_tops_bench_$c.err; tail -c 600 gpurun_out/r05_tops_bench_$c.json | cut -c1-300; done
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 | tee gpurun_out/r05_gpu_tests.log
