#!/bin/bash
# round 5: kernel stats of the y-solve with dense tree tops: bash tools/gpu_r5_tops_prof.sh <fixture> <switch_admm> <iters> [key=value ...]
: "${GRAFT_REPO_ROOT:?}"
set -u
R="$GRAFT_REPO_ROOT"
N="$1"; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p "$R/gpurun_out"
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_t" -- python3 "$R/tools/run_large.py" "$N" "$@" > "$R/gpurun_out/r05_tops_${N}_run.log" 2>&1
cp "$(find "$R/gpurun_out/prof_t" -name '*kernel_stats.csv' | head -1)" "$R/gpurun_out/r05_tops_${N}_kernel_stats.csv"
rm -rf "$R/gpurun_out/prof_t"
head -24 "$R/gpurun_out/r05_tops_${N}_kernel_stats.csv" | cut -c1-200
grep RESULT "$R/gpurun_out/r05_tops_${N}_run.log" | cut -c1-400
