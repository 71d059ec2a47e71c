#!/bin/bash
# Round-5 profiling on the GPU box: per config kernel trace + stats, then PMC passes on their own (no trace domains with --pmc).
#   bash tools/prof_round5.sh [configs...]      (default: c2 c2_20 c4 c1 c5 c3)
: "${GRAFT_REPO_ROOT:?}"
set -u
R="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
G="$R/gpurun_out"
mkdir -p "$G"
prof () {   # tag, command...
  tag=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$G/prof_${tag}" -- "$@" > "$G/${tag}_trace.log" 2>&1
  timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$G/prof_${tag}_fetch" -- "$@" > "$G/${tag}_fetch.log" 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$G/prof_${tag}_write" -- "$@" > "$G/${tag}_write.log" 2>&1
  timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d "$G/prof_${tag}_sq" -- "$@" > "$G/${tag}_sq.log" 2>&1
  # the instruction mix of the kernels (what the SIMDs issue beside the MFMAs: tools/ubench/mfma_coissue.hip), two more passes
  timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_INSTS_BRANCH --output-format csv -d "$G/prof_${tag}_sq2" -- "$@" > "$G/${tag}_sq2.log" 2>&1
  timeout 900 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64 SQ_INSTS --output-format csv -d "$G/prof_${tag}_sq3" -- "$@" > "$G/${tag}_sq3.log" 2>&1
  grep '^{' "$G/${tag}_fetch.log" > "$G/r05_${tag}_bench_under_rocprof.json" || true
  python3 "$R/tools/summarize_prof.py" "r05_${tag}" "$(dirname "$(find "$G/prof_${tag}" -name "*kernel_stats.csv" | head -1)")" \
      "$(dirname "$(find "$G/prof_${tag}_fetch" -name "*counter_collection.csv" | head -1)")" \
      "$(dirname "$(find "$G/prof_${tag}_write" -name "*counter_collection.csv" | head -1)")" \
      "$(dirname "$(find "$G/prof_${tag}_sq" -name "*counter_collection.csv" | head -1)"):$(dirname "$(find "$G/prof_${tag}_sq2" -name "*counter_collection.csv" | head -1)"):$(dirname "$(find "$G/prof_${tag}_sq3" -name "*counter_collection.csv" | head -1)")" "$G/r05_${tag}_bench_under_rocprof.json" > "$G/${tag}_summary.log" 2>&1
  rm -rf "$G/prof_${tag}" "$G/prof_${tag}_fetch" "$G/prof_${tag}_write" "$G/prof_${tag}_sq" "$G/prof_${tag}_sq2" "$G/prof_${tag}_sq3"
}
CFG="${*:-c2 c2_20 c4 c1 c5 c3}"
for c in $CFG; do
  case $c in
    c2)    prof c2 python3 "$R/bench.py" --steps 200 --warmup 20 --no-cpu-baseline --no-breakdown ;;
    c2_20) prof c2_20 python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-breakdown ;;
    c3)    prof c3 python3 "$R/bench.py" --config c3 --steps 10 --warmup 2 --no-cpu-baseline --no-breakdown ;;
    c4)    prof c4 python3 "$R/bench.py" --config c4 --steps 40 --warmup 5 --no-cpu-baseline --no-breakdown ;;
    c1)    prof c1 python3 "$R/bench.py" --config c1 --steps 60 --warmup 5 --no-cpu-baseline --no-breakdown --time-to-tol 0 ;;
    c5)    prof c5 python3 "$R/bench.py" --config c5 --steps 100 --warmup 5 --no-cpu-baseline --no-breakdown --time-to-tol 0 ;;
  esac
done
cd "$R"
cp profiles/r05_*_kernel_stats.csv profiles/r05_*_pmc_*.json gpurun_out/ 2>/dev/null
ls profiles | grep r05
