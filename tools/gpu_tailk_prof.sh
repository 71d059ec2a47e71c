#!/bin/bash
# per-kernel times of the device-side y-solve at several forced tail sizes -> gpurun_out/tailk_prof.log
: "${GRAFT_REPO_ROOT:?}"
R="$GRAFT_REPO_ROOT"
mkdir -p $R/gpurun_out
O=$R/gpurun_out/tailk_prof.log
: > $O
cd /tmp && export TMPDIR=/tmp
for K in 16896 17408 17920 18432 20480; do
  rm -rf /tmp/tkp_$K
  CUADMM_TAIL_K=$K rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tkp_$K -o p -- python3 $R/tools/run_large.py PushBox_N=30_MOMENT 11000 400 lead_debug=1 2>&1 | grep -E "RESULT|lead debug" | sed "s/^/[k=$K] /" >> $O
  F=$(find /tmp/tkp_$K -name "*kernel_stats.csv" | head -1)
  echo "[k=$K] kernel stats:" >> $O
  python3 -c "import csv,sys; [print('   %-60s calls %6s avg_us %9.1f pct %5s' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage'])) for i, r in enumerate(csv.DictReader(open(sys.argv[1]))) if i < 12]" "$F" >> $O
done
cat $O
