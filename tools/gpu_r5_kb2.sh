#!/bin/bash
# round 5: lead_small_kb chosen at build against 16, alternating, on the bench lines of the coupled configurations
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
  for kb in 16 0; do
    for c in c1 c5; do
      timeout 400 python bench.py --config $c --steps 400 --warmup 40 --no-cpu-baseline --no-breakdown --time-to-tol 0 --option lead_small_kb=$kb 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$c kb=$kb', round(d['value'],1), round(d['steady_state']['value'],1))"
    done
  done
done
