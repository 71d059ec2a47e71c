#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
./tools/ubench/mfma_4x4_layout.exe > gpurun_out/r05_mfma_4x4_layout.log 2>&1
head -40 gpurun_out/r05_mfma_4x4_layout.log
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "tail_solve" 2>&1 | tail -5
