#!/bin/bash
# round 5: the schedule's warm start on single mid-size matrices (N > 512): psd_hint = 2 against the default
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
run() { timeout 900 python tools/run_large.py "$@" 2>&1 | grep "RESULT" | grep -o "RESULT [A-Za-z_=0-9.-]*\|'psd_hint': [0-9.]*\|[0-9.]* ms/iter\|'psd_project': np.float64([0-9.]*)" | paste - - - -; }
for h in 1 2; do
  run 1dc.1024 11000 353 psd_hint=$h
  run swissroll 11000 600 psd_hint=$h
  run bqp-r1-40-1 11000 1000 psd_hint=$h
done
