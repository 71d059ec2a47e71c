#!/bin/bash
# ONE runner for everything that goes to the GPU box (replaces the 72 one-off gpu_r*.sh / prof_round*.sh of rounds 1 - 5):
#
#     gpurun --timeout 1500 -- 'bash tools/gpurun.sh <recipe> [args]'          results under gpurun_out/, tagged r${ROUND}
#
#   suite                      the whole `-m gpu` suite with --durations=25             -> gpurun_out/r${ROUND}_gpu_tests.log
#   long                       what the default suite leaves out for its time budget (CUADMM_LONG_TESTS=1, 23 soak rounds)
#   ranks                      the multi-rank / transport tests only (RCCL at world 1; 2, 4, 8 ranks on the one GPU)
#   smoke                      __graft_entry__.smoke() + the driver's bench line
#   bench [cfg ...]            bench lines (default: c2 at the driver's settings, c2 c3 c4 c1 c5 at their defaults) -> r${ROUND}_bench_<cfg>.json
#   ab <key> <v1,v2,..> [cfg ...]   the same bench lines under engine option key=value, one line per value and config
#   prof [cfg ...]             rocprofv3: kernel trace + stats, then every PMC pass ON ITS OWN -> profiles-ready summaries r${ROUND}_<cfg>_*
#   trace <tag> <bench args>   kernel stats of one bench configuration
#   real [fixture switch cap [k=v ...]]   the reference's largest shipped inputs to 1e-3 (tools/run_large.py); no args: all of them
#   moment "<keys>" "<k=v,..;k=v,..>"   oracle-trajectory deviations under option variants (tools/probe_moment.py)
#   convergence [cap]          BASELINE configs[4] with the reference log's parameters (bench.py --config c5 --convergence-cap)
#   ubench <name> [args]       build tools/ubench/<name>.hip for gfx950 and run it
set -u
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"
R="$GRAFT_REPO_ROOT"
ROUND="${ROUND:-06}"
G="$R/gpurun_out"
mkdir -p "$G"
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "$R"
recipe="${1:-suite}"; shift || true

bench_args () {   # the arguments of one configuration's line
  case $1 in
    c2_20) echo "--steps 20 --warmup 5" ;;
    c2)    echo "--steps 200 --warmup 20" ;;
    *)     echo "--config $1" ;;
  esac
}

prof () {   # tag, command...: kernel trace + stats, then the counter passes on their own (never --pmc beside a trace domain)
  tag=$1; shift
  # the one-time factorisation of the y-solve's dense tail launches two to three short kernels per COLUMN (pivot search, swap, column: ~30 000
  # dispatches at init on c1); rocprofv3's counter collection dies (SIGSEGV inside the profiler) somewhere past that many instrumented
  # dispatches, so the init-only kernels are left out of the PMC passes -- the per-iteration kernels are what the summaries are about
  X='--kernel-exclude-regex ts_piv_|ts_ldlt_|ts_gemm_|ts_diag_inverse|ts_scatter_csr|ts_transpose'
  ( cd /tmp && export TMPDIR=/tmp
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$G/prof_${tag}" -- "$@" > "$G/${tag}_trace.log" 2>&1
    timeout 900 rocprofv3 --pmc FETCH_SIZE $X --output-format csv -d "$G/prof_${tag}_fetch" -- "$@" > "$G/${tag}_fetch.log" 2>&1
    timeout 900 rocprofv3 --pmc WRITE_SIZE $X --output-format csv -d "$G/prof_${tag}_write" -- "$@" > "$G/${tag}_write.log" 2>&1
    timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES $X --output-format csv -d "$G/prof_${tag}_sq" -- "$@" > "$G/${tag}_sq.log" 2>&1
    timeout 900 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_INSTS_BRANCH $X --output-format csv -d "$G/prof_${tag}_sq2" -- "$@" > "$G/${tag}_sq2.log" 2>&1 )
  grep '^{' "$G/${tag}_fetch.log" > "$G/r${ROUND}_${tag}_bench_under_rocprof.json" || true
  d () { dirname "$(find "$1" -name "$2" | head -1)"; }
  python3 "$R/tools/summarize_prof.py" "r${ROUND}_${tag}" "$(d "$G/prof_${tag}" '*kernel_stats.csv')" "$(d "$G/prof_${tag}_fetch" '*counter_collection.csv')" \
      "$(d "$G/prof_${tag}_write" '*counter_collection.csv')" "$(d "$G/prof_${tag}_sq" '*counter_collection.csv'):$(d "$G/prof_${tag}_sq2" '*counter_collection.csv')" \
      "$G/r${ROUND}_${tag}_bench_under_rocprof.json" > "$G/${tag}_summary.log" 2>&1
  rm -rf "$G/prof_${tag}" "$G/prof_${tag}_fetch" "$G/prof_${tag}_write" "$G/prof_${tag}_sq" "$G/prof_${tag}_sq2"
}

case "$recipe" in
  suite)
    python -m pytest tests -m gpu -q --durations=25 > "$G/r${ROUND}_gpu_tests.log" 2>&1; tail -40 "$G/r${ROUND}_gpu_tests.log" ;;
  long)
    CUADMM_LONG_TESTS=1 CUADMM_SOAK_ROUNDS=23 python -m pytest tests/test_gpu_moment_parity.py tests/test_gpu_soak_cluster.py -m gpu -q --durations=5 \
      -k "pusht30_with_the_factor or soak" > "$G/r${ROUND}_gpu_tests_long.log" 2>&1; tail -12 "$G/r${ROUND}_gpu_tests_long.log" ;;
  ranks)
    python -m pytest tests/test_gpu_rccl.py tests/test_gpu_sharded_procs.py tests/test_gpu_bench_ranks.py tests/test_gpu_sharded.py -m gpu -q --durations=20 \
      > "$G/r${ROUND}_gpu_tests_ranks.log" 2>&1; tail -30 "$G/r${ROUND}_gpu_tests_ranks.log" ;;
  smoke)
    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
    python bench.py --steps 20 --warmup 5 | tee "$G/r${ROUND}_bench_driver.json" | cut -c1-400 ;;
  bench)
    for c in ${*:-c2_20 c2 c3 c4 c1 c5}; do
      python bench.py $(bench_args $c) > "$G/r${ROUND}_bench_$c.json" 2> "$G/r${ROUND}_bench_$c.err"
      python - "$G/r${ROUND}_bench_$c.json" $c <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "value %.1f" % d["value"], "ms %.4f" % d["ms_per_step"], "frac %.4f" % d["roofline"]["frac"], "steady %.1f" % d["steady_state"]["value"],
      "y_solve", (d.get("y_solve") or {}).get("ms_per_iteration"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
P
    done ;;
  ab)
    key=$1; vals=$2; shift 2
    for v in ${vals//,/ }; do for c in ${*:-c1 c5}; do
      python bench.py $(bench_args $c) --no-cpu-baseline --time-to-tol 0 --option $key=$v > "$G/r${ROUND}_ab_${key}_${v}_$c.json" 2> "$G/r${ROUND}_ab_${key}_${v}_$c.err"
      python - "$G/r${ROUND}_ab_${key}_${v}_$c.json" "$key=$v" $c <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], sys.argv[3], "value %.1f" % d["value"], "steady %.1f" % d["steady_state"]["value"], "y_solve", (d.get("y_solve") or {}).get("ms_per_iteration"),
      {k: round(v, 4) for k, v in d["breakdown_ms_per_iter"].items()})
P
    done; done ;;
  abm)   # abm "k=v k=v;k=v k=v" [cfg ...]: the same lines under SETS of engine options
    sets=$1; shift
    IFS=';' read -ra SETS <<< "$sets"
    for set_ in "${SETS[@]}"; do for c in ${*:-c1 c5}; do
      tag=$(echo "$set_" | tr ' =' '__'); opts=""
      for kv in $set_; do opts="$opts --option $kv"; done
      python bench.py $(bench_args $c) --no-cpu-baseline --time-to-tol 0 $opts > "$G/r${ROUND}_ab_${tag}_$c.json" 2> "$G/r${ROUND}_ab_${tag}_$c.err"
      python - "$G/r${ROUND}_ab_${tag}_$c.json" "$set_" $c <<'P'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], "|", sys.argv[3], "value %.1f" % d["value"], "steady %.1f" % d["steady_state"]["value"], "y_solve", (d.get("y_solve") or {}).get("ms_per_iteration"))
P
    done; done ;;
  prof)
    for c in ${*:-c2 c2_20 c4 c1 c5 c3}; do
      case $c in
        c2)    prof c2 python3 "$R/bench.py" --steps 200 --warmup 20 --no-cpu-baseline --no-breakdown ;;
        c2_20) prof c2_20 python3 "$R/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-breakdown ;;
        c3)    prof c3 python3 "$R/bench.py" --config c3 --steps 10 --warmup 2 --no-cpu-baseline --no-breakdown ;;
        c4)    prof c4 python3 "$R/bench.py" --config c4 --steps 40 --warmup 5 --no-cpu-baseline --no-breakdown ;;
        c1)    prof c1 python3 "$R/bench.py" --config c1 --steps 60 --warmup 5 --no-cpu-baseline --no-breakdown --time-to-tol 0 ;;
        c5)    prof c5 python3 "$R/bench.py" --config c5 --steps 100 --warmup 5 --no-cpu-baseline --no-breakdown --time-to-tol 0 ;;
      esac
    done
    # the summaries were written under profiles/ of the box's copy: only gpurun_out/ travels back
    cp "$R"/profiles/r${ROUND}_*_kernel_stats.csv "$R"/profiles/r${ROUND}_*_pmc_*.json "$G"/ 2>/dev/null
    ls "$G" | grep "r${ROUND}_.*\(kernel_stats\|pmc_\)" ;;
  trace)
    tag=$1; shift
    ( cd /tmp && export TMPDIR=/tmp
      timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$G/prof_$tag" -- python3 "$R/bench.py" "$@" --no-cpu-baseline --no-breakdown > "$G/${tag}_trace.log" 2>&1 )
    cp "$(find "$G/prof_$tag" -name '*kernel_stats.csv' | head -1)" "$G/r${ROUND}_${tag}_kernel_stats.csv"; rm -rf "$G/prof_$tag"
    head -24 "$G/r${ROUND}_${tag}_kernel_stats.csv" | cut -c1-200 ;;
  real)
    O="$G/r${ROUND}_real_data.log"
    if [ $# -gt 0 ]; then timeout 1500 python tools/run_large.py "$@" 2>&1 | grep RESULT | tee -a "$O" | cut -c1-420
    else
      : > "$O"
      for a in "PushBox_N=30_MOMENT 11000 60000" "PushBox_N=30_MOMENT 0 60000" "PushBox_N=50_MOMENT 11000 60000" "PlanarHand_N=10_MOMENT 11000 25000" "PushT_N=30_MOMENT 11000 60000"; do
        timeout 1500 python tools/run_large.py $a 2>&1 | grep RESULT >> "$O"
      done
      cut -c1-420 "$O"
    fi ;;
  moment)
    python tools/probe_moment.py "$1" "${2:-}" 2>&1 | tee -a "$G/r${ROUND}_moment_probe.log" | grep -v absdiff ;;
  convergence)
    python bench.py --config c5 --no-cpu-baseline --convergence-cap "${1:-1000000}" > "$G/r${ROUND}_pendulum_convergence.json" 2> "$G/r${ROUND}_pendulum_convergence.err"
    python -c "import json; print(json.dumps(json.load(open('$G/r${ROUND}_pendulum_convergence.json'))['time_to_tol'], indent=1))" ;;
  ubench)
    n=$1; shift
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I"$R/include" -I"$R/cuadmm_amd/csrc" "tools/ubench/$n.hip" -o "tools/ubench/$n.exe" && "tools/ubench/$n.exe" "$@" 2>&1 | tee "$G/r${ROUND}_ubench_$n.log" ;;
  *) echo "unknown recipe $recipe"; exit 2 ;;
esac
