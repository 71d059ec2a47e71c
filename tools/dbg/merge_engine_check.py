"""Engine trajectories with merged one-launch groups against the per-group launches on a synthetic problem with the given blocks.
python tools/dbg/merge_engine_check.py 252 56 56 56 126"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import cuadmm_amd  # noqa: E402
from cuadmm_amd import synthetic  # noqa: E402

blk = [int(a) for a in sys.argv[1:]]
prob = synthetic.make_synthetic(blk, cons_per_block=5, dense_C=True)


def run(opts):
    s = cuadmm_amd.SDPSolver(verbose=False, options=opts)
    s.init_problem(cuadmm_amd.Problem(prob.vec_len, prob.con_num, prob.blk, prob.At_col_ptrs, prob.At_row_ids, prob.At_vals, prob.b_idx, prob.b_vals, prob.C_idx, prob.C_vals))
    s.solve(80, 0.0, 0, 50, 100, 11000, 1.05)
    return np.array(s.info_arr("pobj")), np.array(s.info_arr("errRd")), s.counters()


extra = {}
for kv in os.environ.get("EXTRA", "").split():
    k, v = kv.split("=")
    extra[k] = float(v)
ref = run(dict(extra, psd_lg_merge=0))
same_ref = all(np.array_equal(run(dict(extra, psd_lg_merge=0))[0], ref[0]) for _ in range(3))
res = []
for trial in range(6):
    a = run(dict(extra, psd_lg_merge=1))
    res.append(-1 if np.array_equal(a[0], ref[0]) and np.array_equal(a[1], ref[1]) else int(np.argmax(a[0] != ref[0])))
print(blk, extra, "per-group reproducible:", same_ref, "| merged first-diff iterations (-1 = identical):", res)
