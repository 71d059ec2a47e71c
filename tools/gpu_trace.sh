#!/bin/bash
# kernel trace + stats of one bench configuration: bash tools/gpu_trace.sh <tag> <bench args...>
: "${GRAFT_REPO_ROOT:?}"
set -u
R="$GRAFT_REPO_ROOT"; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_$tag" -- python3 "$R/bench.py" "$@" --no-cpu-baseline --no-breakdown > "$R/gpurun_out/${tag}_trace.log" 2>&1
cp "$(find "$R/gpurun_out/prof_$tag" -name '*kernel_stats.csv' | head -1)" "$R/gpurun_out/${tag}_kernel_stats.csv"
rm -rf "$R/gpurun_out/prof_$tag"
python3 - "$R/gpurun_out/${tag}_kernel_stats.csv" <<'P'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print("%-100s calls %6s total_ms %9.3f avg_us %9.2f pct %s" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
P
