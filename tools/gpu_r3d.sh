#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
G=$R/gpurun_out
mkdir -p $G
rocprofv3 -L 2>/dev/null | grep -oE "(SQ|TCP|TCC|TA|TD|GRBM)_[A-Z0-9_a-z]+" | sort -u > $G/r3d_counters.txt
wc -l $G/r3d_counters.txt
grep -E "LEVEL|LATENCY|IFETCH|WAIT" $G/r3d_counters.txt | tr '\n' ' '
pm () { tag=$1; shift; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $G/prof_r3d_$tag -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-breakdown --batch 100 > $G/r3d_$tag.log 2>&1; }
pm sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR
pm sq2 SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
pm tcp TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
pm tcc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_HIT_sum TCC_MISS_sum
python3 - <<'P'
import csv,glob,collections,os
G=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out"
for tag in ("sq1","sq2","tcp","tcc"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(G+"/prof_r3d_%s/**/*counter_collection.csv"%tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if "cu_kernel" in r["Kernel_Name"] or "closed_kernel" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        print(tag,k,{c:(sum(x)/len(x),len(x)) for c,x in v.items()})
P
