#!/bin/bash
# PMC passes over the 4x4x4 prototype's kernels (counters only: no trace domains with --pmc)
: "${GRAFT_REPO_ROOT:?}"
set -u
R="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
G="$R/gpurun_out"; mkdir -p "$G"
EXE="$R/tools/ubench/sym4x4_proto.exe"
pass () { tag=$1; shift; timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$G/proto_$tag" -- "$EXE" 4 2 > "$G/proto_$tag.log" 2>&1; }
pass p1 SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
pass p2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS
pass p3 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_SCA
python3 - "$G" <<'PY'
import sys, glob, csv, collections
G = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(G + "/proto_p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"][:28]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(G + "/r05_sym4x4_proto_pmc.txt", "w") as out:
    for k in sorted(acc):
        out.write(k + "\n")
        for c in sorted(acc[k]):
            v = acc[k][c]
            out.write("   %-28s %.4g (mean of %d dispatches)\n" % (c, sum(v) / len(v), len(v)))
print(open(G + "/r05_sym4x4_proto_pmc.txt").read())
PY
rm -rf "$G"/proto_p1 "$G"/proto_p2 "$G"/proto_p3
