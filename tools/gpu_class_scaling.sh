#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
set -u
R="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/prof_cs" -- python3 "$R/tools/probe_class_scaling.py" "$@" > "$R/gpurun_out/cs.log" 2>&1
python3 - "$(find "$R/gpurun_out/prof_cs" -name '*kernel_trace.csv' | head -1)" <<'P'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "psd_sign_wave_kernel" in r["Kernel_Name"] or "psd_sign_lds" in r["Kernel_Name"]]
for r in rows:
    print("%-60s grid %8s  %9.1f us" % (r["Kernel_Name"][:60], r.get("Grid_Size_X", r.get("Grid_Size","?")), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
P
rm -rf "$R/gpurun_out/prof_cs"
