"""Developer aid: time of solve(K, if_first=False) on C2 as a function of K (fixed cost per solve call, per launch, per iteration)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cuadmm_amd
from cuadmm_amd import synthetic
lib = cuadmm_amd.load()
prob = synthetic.config_c2(10000, 32)
opts = {}
for kv in sys.argv[1:]:
    k, v = kv.split("="); opts[k] = float(v)
s = cuadmm_amd.SDPSolver(verbose=False, options=opts)
s.init_problem(cuadmm_amd.Problem(prob.vec_len, prob.con_num, prob.blk, prob.At_col_ptrs, prob.At_row_ids, prob.At_vals, prob.b_idx, prob.b_vals, prob.C_idx, prob.C_vals))
s.solve(40, 0.0, 0, 50, 100, 0, 1.05)
for _ in range(3):
    s.solve(20, 0.0, 0, 50, 100, 0, 1.05, if_first=False)
lib.cuadmm_dev_sync()
res = {}
for K in (1, 2, 3, 5, 10, 20, 40, 64, 65, 100):
    ts = []
    for rep in range(5):
        lib.cuadmm_dev_sync()
        t0 = time.perf_counter()
        s.solve(K, 0.0, 0, 50, 100, 0, 1.05, if_first=False)
        lib.cuadmm_dev_sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    res[K] = min(ts)
    print("K %3d  min %.3f ms  median %.3f  per-iter %.4f" % (K, min(ts), sorted(ts)[2], min(ts) / K), flush=True)
print("opts", opts, "20-step iters/s %.0f" % (20 / res[20] * 1e3), "counters", s.counters())
