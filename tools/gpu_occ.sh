cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for pad in 0 9000 29000; do
  export CUADMM_PSD_W32_PAD=$pad
  CUADMM_PSD_DEBUG=1 python3 tools/probe_w32_occ.py 32 10000 1 2>&1 | grep -v "^$" | tail -3
  rocprofv3 --kernel-trace --stats -d /tmp/occ_$pad -o occ -- python3 tools/probe_w32_occ.py 32 10000 5 > /dev/null 2>&1
  f=$(find /tmp/occ_$pad -name "*kernel_stats.csv" | head -1)
  echo "pad $pad: $(grep wave32 $f | cut -d, -f1-5)"
done
