"""Per-call time of consecutive 20-iteration solves right after init (clock ramp / warm-up of the engine)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuadmm_amd
from cuadmm_amd import synthetic
lib = cuadmm_amd.load()
prob = synthetic.config_c2(10000, 32)
s = cuadmm_amd.SDPSolver(verbose=False, options={"batch": int(sys.argv[1]) if len(sys.argv) > 1 else 64})
s.init_problem(cuadmm_amd.Problem(prob.vec_len, prob.con_num, prob.blk, prob.At_col_ptrs, prob.At_row_ids, prob.At_vals, prob.b_idx, prob.b_vals, prob.C_idx, prob.C_vals))
t_all = time.perf_counter()
s.solve(5, 0.0, 0, 50, 100, 0, 1.05)
lib.cuadmm_dev_sync()
out = []
for i in range(12):
    t0 = time.perf_counter()
    s.solve(20, 0.0, 0, 50, 100, 0, 1.05, if_first=False)
    lib.cuadmm_dev_sync()
    out.append((time.perf_counter() - t0) * 1e3 / 20)
print("ms/iter per 20-iteration call:", " ".join("%.4f" % x for x in out), "| steps", float(s.psd_steps().mean()) if False else "")

# the same on a FRESH solver after 150 ms of unrelated matrix-core work (clock ramp or engine warm-up?)
import numpy as np, ctypes as C
from tests.helpers import Dev
from cuadmm_amd._lib import check
blk = np.full(10000, 32, np.int32)
x = np.random.default_rng(0).standard_normal(10000 * 528)
din, dout = Dev(x), Dev(shape=(x.size,), dtype=np.float64)
s2 = cuadmm_amd.SDPSolver(verbose=False, options={"batch": int(sys.argv[1]) if len(sys.argv) > 1 else 64})
s2.init_problem(cuadmm_amd.Problem(prob.vec_len, prob.con_num, prob.blk, prob.At_col_ptrs, prob.At_row_ids, prob.At_vals, prob.b_idx, prob.b_vals, prob.C_idx, prob.C_vals))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15:
    check(lib.cuadmm_op_psd_project(din.ptr, dout.ptr, blk.ctypes.data_as(C.c_void_p), int(blk.size), None))
s2.solve(5, 0.0, 0, 50, 100, 0, 1.05)
lib.cuadmm_dev_sync()
out = []
for i in range(8):
    t0 = time.perf_counter()
    s2.solve(20, 0.0, 0, 50, 100, 0, 1.05, if_first=False)
    lib.cuadmm_dev_sync()
    out.append((time.perf_counter() - t0) * 1e3 / 20)
print("after 150 ms of projections: ", " ".join("%.4f" % x for x in out))
