"""Developer aid: per-call time of consecutive 20-iteration solves right after init (warm-up of the engine: schedule hints, longest block
first, the iterates themselves).  python tools/probe_rampup.py [key=value ...]"""
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
s = cuadmm_amd.SDPSolver(verbose=False, psd_steps=True, options=opts)
s.init_problem(cuadmm_amd.Problem(prob.vec_len, prob.con_num, prob.blk, prob.At_col_ptrs, prob.At_row_ids, prob.At_vals, prob.b_idx, prob.b_vals, prob.C_idx, prob.C_vals))
s.solve(5, 0.0, 0, 50, 100, 0, 1.05)
lib.cuadmm_dev_sync()
out = []
for i in range(12):
    t0 = time.perf_counter()
    s.solve(20, 0.0, 0, 50, 100, 0, 1.05, if_first=False)
    lib.cuadmm_dev_sync()
    dt = (time.perf_counter() - t0) * 1e3 / 20
    st = s.psd_steps()
    out.append("%.4f(%.2f/%d)" % (dt, st.mean(), st.max()))
print(opts, "ms/iter per 20-iteration call (steps mean/max of its last iteration):", " ".join(out))

# the same on a second solver that was initialised BEFORE a dense 130 ms of GPU work by the first one: engine warm-up or clock ramp?
s2 = cuadmm_amd.SDPSolver(verbose=False, psd_steps=True, options=opts)
s2.init_problem(cuadmm_amd.Problem(prob.vec_len, prob.con_num, prob.blk, prob.At_col_ptrs, prob.At_row_ids, prob.At_vals, prob.b_idx, prob.b_vals, prob.C_idx, prob.C_vals))
s.solve(600, 0.0, 0, 50, 100, 0, 1.05, if_first=False)
s2.solve(5, 0.0, 0, 50, 100, 0, 1.05)
lib.cuadmm_dev_sync()
out = []
for i in range(8):
    t0 = time.perf_counter()
    s2.solve(20, 0.0, 0, 50, 100, 0, 1.05, if_first=False)
    lib.cuadmm_dev_sync()
    out.append("%.4f" % ((time.perf_counter() - t0) * 1e3 / 20))
print("fresh solver right after 130 ms of dense work:", " ".join(out))
