#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
./tools/ubench/sym4x4_proto.exe 8 4 > gpurun_out/r05_sym4x4_proto.log 2>&1
cat gpurun_out/r05_sym4x4_proto.log
