#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_psd.py -q -m gpu -x -k "batch_eig" -s 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_eig" -- python3 "$GRAFT_REPO_ROOT/tools/probe_eig_large.py" 2000 > "$GRAFT_REPO_ROOT/gpurun_out/eig_trace.log" 2>&1
cd "$GRAFT_REPO_ROOT"
cp "$(find gpurun_out/prof_eig -name '*kernel_stats.csv' | head -1)" gpurun_out/r03_eig2000_kernel_stats.csv
rm -rf gpurun_out/prof_eig
head -25 gpurun_out/r03_eig2000_kernel_stats.csv | cut -c1-200
tail -4 gpurun_out/eig_trace.log
