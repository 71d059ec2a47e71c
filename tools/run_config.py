"""Times BASELINE configs on the engine: c2 | c3 [n] | c4 [n_blocks].  Prints ms/iter and the per-class profile."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cuadmm_amd
from cuadmm_amd import synthetic
which = sys.argv[1]; arg = int(sys.argv[2]) if len(sys.argv) > 2 else None; iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
p = {"c2": lambda: synthetic.config_c2(arg or 10000), "c3": lambda: synthetic.config_c3(arg or 2000),
     "c4": lambda: synthetic.config_c4(arg or 100000)}[which]()
prob = cuadmm_amd.Problem(p.vec_len, p.con_num, p.blk, p.At_col_ptrs, p.At_row_ids, p.At_vals, p.b_idx, p.b_vals, p.C_idx, p.C_vals)
s = cuadmm_amd.SDPSolver(verbose=False, profile=1)
t = time.time(); s.init_problem(prob); ti = time.time() - t
s.solve(10, 0.0, 0, 50, 100, 0, 1.05)           # warm-up
s.reset_profile() if hasattr(s, "reset_profile") else None
t = time.time(); s.solve(iters, 0.0, 0, 50, 100, 0, 1.05); ts = time.time() - t
n_it = s.info_iter_num
print("RESULT %s blocks %d m %d L %d: init %.2fs, %d it in %.2fs -> %.3f ms/iter" % (which, p.blk.size, p.con_num, p.vec_len, ti, n_it, ts, ts / max(n_it, 1) * 1e3))
for k, v in s.profile().items():
    if v["launches"]: print("   %-12s per-iter %.3f ms" % (k, v["ms"] / max(n_it, 1)))
st = s.state(); print({k: st[k] for k in ("errRp", "errRd", "relgap", "pobj", "dobj")})
