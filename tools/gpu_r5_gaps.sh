#!/bin/bash
# where an iteration of a latency-bound problem spends its time BETWEEN kernels: kernel trace of bench.py --config cX, idle gaps per iteration
: "${GRAFT_REPO_ROOT:?}"
set -u
R="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
for c in c5 c1; do
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/prof_gap" -- python3 "$R/bench.py" --config $c --steps 60 --warmup 10 --no-cpu-baseline --no-breakdown > "$R/gpurun_out/gap_trace.log" 2>&1
python3 - "$(find "$R/gpurun_out/prof_gap" -name '*kernel_trace.csv' | head -1)" $c <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# take a window in the middle of the run; iterations are delimited by the first kernel of an iteration (aty_xb* or lead_sweep forward...)
n=len(rows); lo=n//3; hi=lo+ (n//6)
win=rows[lo:hi]
t_first=int(win[0]["Start_Timestamp"]); t_last=max(int(r["End_Timestamp"]) for r in win)
busy=0; cur_end=t_first; gaps=[]
for r in win:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if s>cur_end: gaps.append((s-cur_end, r["Kernel_Name"][:60]))
    if e>cur_end:
        busy+= e-max(s,cur_end); cur_end=e
span=t_last-t_first
print(sys.argv[2],"window: %d kernels, span %.1f us, GPU busy (union) %.1f us = %.1f%%, idle %.1f us" % (len(win), span/1e3, busy/1e3, 100*busy/span, (span-busy)/1e3))
import collections
g=collections.defaultdict(lambda:[0,0])
for d,k in gaps: g[k][0]+=d; g[k][1]+=1
for k,(d,cn) in sorted(g.items(), key=lambda kv:-kv[1][0])[:8]:
    print("   idle before %-62s total %8.1f us over %4d gaps (%.1f us each)" % (k, d/1e3, cn, d/1e3/cn))
P
rm -rf "$R/gpurun_out/prof_gap"
done
