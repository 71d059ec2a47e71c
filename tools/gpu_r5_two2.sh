#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python bench.py --gpus 2 --blocks-per-gpu 5000 --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/two.out 2> gpurun_out/two.err
echo "rc=$?"; tail -c 1500 gpurun_out/two.err; head -c 600 gpurun_out/two.out
