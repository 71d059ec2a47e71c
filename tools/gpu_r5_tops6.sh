#!/bin/bash
# round 5: kernel stats of the small moment problems with dense tree tops (calibration of plan_tail's model for small forests)
: "${GRAFT_REPO_ROOT:?}"
set -u
R="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
mkdir -p "$R/gpurun_out"
prof() {
  tag="$1"; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_t" -- python3 "$R/tools/run_large.py" "$@" > "$R/gpurun_out/r05_tops_${tag}_run.log" 2>&1
  cp "$(find "$R/gpurun_out/prof_t" -name '*kernel_stats.csv' | head -1)" "$R/gpurun_out/r05_tops_${tag}_kernel_stats.csv"
  rm -rf "$R/gpurun_out/prof_t"
  grep RESULT "$R/gpurun_out/r05_tops_${tag}_run.log" | cut -c1-200
  python3 - "$R/gpurun_out/r05_tops_${tag}_kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:40]:
    n = r["Name"].replace("cuadmm::", "").replace("(anonymous namespace)::", "")
    if any(k in n for k in ("lead_", "tops_", "ts_onepass", "ts_tri")): print("   %-50s calls %5s avg_us %8.1f" % (n[:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
}
prof pend_default pendulum_N=80 11000 300
prof pend_6144_16 pendulum_N=80 11000 300 tail_k=6144 lead_tops=16
prof pend_6144_32 pendulum_N=80 11000 300 tail_k=6144 lead_tops=32
prof pend_4096_32 pendulum_N=80 11000 300 tail_k=4096 lead_tops=32
prof ph1_default PlanarHand_N=1_MOMENT 0 300
prof pb30_default PushBox_N=30_MOMENT 11000 300
