#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
G=$R/gpurun_out
mkdir -p $G
pm () { tag=$1; shift; timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $G/prof_r3g_$tag -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-breakdown --batch 0 > $G/r3g_$tag.log 2>&1; }
pm i1 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES
pm i2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES
pm i3 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU
python3 - <<'P'
import csv,glob,collections,os
G=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out"
for tag in ("i1","i2","i3"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(G+"/prof_r3g_%s/**/*counter_collection.csv"%tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if "closed_kernel" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        print(tag,k,{c:round(sum(x)/len(x)) for c,x in v.items()}, len(list(v.values())[0]))
P
