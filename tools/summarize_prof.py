"""Condenses rocprofv3 output (gpurun_out/prof_*) into the small summaries committed under profiles/.

    python tools/summarize_prof.py <round-tag> <kernel-trace-dir> <fetch-pmc-dir> <write-pmc-dir>

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of 1024 B; on gfx950 FETCH_SIZE counts
128-byte requests as 64 B for wide coalesced streams, so the read side is doubled (MI355X_MICROARCH.md, HBM).
"""
import collections
import csv
import json
import os
import shutil
import sys

tag, kt, fd, wd = sys.argv[1:5]
sqd = sys.argv[5] if len(sys.argv) > 5 else None
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)
for f in os.listdir(kt):
    if f.endswith("kernel_stats.csv"):
        shutil.copyfile(os.path.join(kt, f), os.path.join(out, "%s_kernel_stats.csv" % tag))


def mean_counter(d, counter):
    agg = collections.defaultdict(list)
    for f in os.listdir(d):
        if f.endswith("counter_collection.csv"):
            for r in csv.DictReader(open(os.path.join(d, f))):
                if r["Counter_Name"] == counter:
                    agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


fetch, write = mean_counter(fd, "FETCH_SIZE"), mean_counter(wd, "WRITE_SIZE")
summary = {}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("void cuadmm") and not k.startswith("cuadmm"):
        continue
    fr = fetch.get(k, (0.0, 0))[0] * 1024.0
    wr = write.get(k, (0.0, 0))[0] * 1024.0
    summary[k] = {"FETCH_SIZE_bytes_raw": fr, "fetch_bytes_corrected_x2": 2.0 * fr, "WRITE_SIZE_bytes": wr,
                  "hbm_bytes_per_launch": 2.0 * fr + wr, "launches_sampled": fetch.get(k, (0, 0))[1]}
# kernels that run SEVERAL ADMM iterations per launch: bytes per iteration from the engine's own counters in the bench line of the
# profiled run (argv[6]: its JSON; engine_plan.batch_iters / batch_launches), so that a reader can compare with another run's launches
bench = None
if len(sys.argv) > 6 and os.path.exists(sys.argv[6]):
    try:
        bench = json.load(open(sys.argv[6]))
    except Exception:
        bench = None
if bench and bench.get("engine_plan", {}).get("batch_launches", 0) > 0:
    ipl = bench["engine_plan"]["batch_iters"] / bench["engine_plan"]["batch_launches"]
    for k, v in summary.items():
        if "closed_cu_kernel" in k:
            v["iterations_per_launch_in_profile"] = ipl
            v["hbm_bytes_per_iteration"] = v["hbm_bytes_per_launch"] / ipl
with open(os.path.join(out, "%s_pmc_hbm_traffic.json" % tag), "w") as f:
    json.dump(summary, f, indent=1)
print(json.dumps(summary, indent=1)[:3000])
if sqd:
    sq = {}
    # every counter found in the SQ pass directories (several directories separated by ':' -- one rocprofv3 run per counter group)
    names = set()
    dirs = [d for d in sqd.split(":") if d and os.path.isdir(d)]
    for d in dirs:
        for f in os.listdir(d):
            if f.endswith("counter_collection.csv"):
                for r in csv.DictReader(open(os.path.join(d, f))):
                    names.add(r["Counter_Name"])
    for d in dirs:
        for counter in sorted(names):
            for k, (v, n) in mean_counter(d, counter).items():
                if k.startswith("void cuadmm") or k.startswith("cuadmm"):
                    sq.setdefault(k, {})[counter] = v
                    sq[k]["launches_sampled"] = n
    with open(os.path.join(out, "%s_pmc_sq_counters.json" % tag), "w") as f:
        json.dump(sq, f, indent=1)
