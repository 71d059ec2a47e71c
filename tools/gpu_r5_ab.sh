#!/bin/bash
: "${GRAFT_REPO_ROOT:?}"
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for opt in "" "--option psd_hint=2" "--option psd_hint=0"; do
for c in c1 c5; do
  timeout 400 python bench.py --config $c --no-cpu-baseline --no-breakdown $opt > gpurun_out/ab.json 2>/dev/null
  python -c "
import json
d=json.load(open('gpurun_out/ab.json'))
print('$c', '$opt', round(d['value'],1), 'steps', d['roofline'].get('newton_schulz_steps'))
"
done
done
python - <<'PY'
import sys, numpy as np
sys.path.insert(0,'.')
import cuadmm_amd
from tests.conftest import load_npz_problem
from tests.helpers import problem_to_amd
p = load_npz_problem("PlanarHand_N=1_MOMENT")
s = cuadmm_amd.SDPSolver(verbose=False, psd_steps=True)
s.init_problem(problem_to_amd(p))
s.solve(120, 0.0, 0, 50, 100, 0, 1.05)
st = s.psd_steps()
blk = np.asarray(p.blk)
big = blk > 16
print("blocks n>16:", blk[big].tolist())
print("steps      :", st[big].tolist())
print("n<=16 with steps>0: mean", st[(~big) & (st > 0)].mean() if ((~big) & (st > 0)).any() else None)
PY
