#!/bin/bash
# rocprofv3 kernel stats of a fixture problem's iterations: bash tools/gpu_stats_problem.sh <name> [iterations]  -> gpurun_out/r03_<name>_kernel_stats.csv
: "${GRAFT_REPO_ROOT:?}"
set -u
R="$GRAFT_REPO_ROOT"
N="$1"; IT="${2:-300}"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_p" -- python3 "$R/tools/probe_problem_breakdown.py" "$N" "$IT" > "$R/gpurun_out/r03_${N}_breakdown.log" 2>&1
cp "$(find "$R/gpurun_out/prof_p" -name '*kernel_stats.csv' | head -1)" "$R/gpurun_out/r03_${N}_kernel_stats.csv"
rm -rf "$R/gpurun_out/prof_p"
head -8 "$R/gpurun_out/r03_${N}_kernel_stats.csv" | cut -c1-160
grep -E "per iteration" "$R/gpurun_out/r03_${N}_breakdown.log"
