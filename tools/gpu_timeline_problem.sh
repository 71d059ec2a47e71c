#!/bin/bash
# kernel timeline of a few iterations of a fixture problem: bash tools/gpu_timeline_problem.sh <name>
: "${GRAFT_REPO_ROOT:?}"
set -u
R="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/prof_tl" -- python3 "$R/tools/probe_problem_breakdown.py" "$1" 60 > "$R/gpurun_out/tl_trace.log" 2>&1
python3 - "$(find "$R/gpurun_out/prof_tl" -name '*kernel_trace.csv' | head -1)" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "cluster" in r["Kernel_Name"] or "lg_gemm" in r["Kernel_Name"]]
c=idx[min(60,len(idx)-1)] if idx else len(rows)//2
t0=int(rows[max(0,c-12)]["Start_Timestamp"])
for r in rows[max(0,c-12):c+24]:
    print("%9.1f %8.1f  q%-3s %s" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Queue_Id","?"), r["Kernel_Name"][:80]))
P
rm -rf "$R/gpurun_out/prof_tl"
