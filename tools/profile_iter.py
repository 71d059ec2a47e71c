"""Per-kernel-class timing of the ADMM iteration on the C2 workload (engine option profile=1)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cuadmm_amd
from cuadmm_amd.synthetic import config_c2
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
mode = sys.argv[3] if len(sys.argv) > 3 else "admm"
t = time.time(); p = config_c2(nb, 32); print("gen %.2fs" % (time.time() - t))
s = cuadmm_amd.SDPSolver(verbose=False, profile=1)
t = time.time()
s.init_problem(cuadmm_amd.Problem(p.vec_len, p.con_num, p.blk, p.At_col_ptrs, p.At_row_ids, p.At_vals, p.b_idx, p.b_vals, p.C_idx, p.C_vals))
print("init %.2fs" % (time.time() - t))
sw = 0 if mode == "admm" else 10 ** 9
s.solve(10, 0.0, 0, 50, 100, sw, 1.05)
s.reset_profile()
t = time.time(); s.solve(steps, 0.0, 0, 50, 100, sw, 1.05, if_first=False); dt = time.time() - t
print("mode %s: %d iters in %.4fs -> %.1f it/s, %.3f ms/it" % (mode, steps, dt, steps / dt, dt / steps * 1e3))
pr = s.profile()
tot = 0
for k, v in pr.items():
    if v["launches"]:
        ms = v["ms"] / v["launches"]
        per_it = v["ms"] / steps
        tot += per_it
        gbs = v["bytes_per_launch"] / (ms * 1e-3) / 1e9 if ms > 0 else 0
        print("  %-12s launches/it %.2f  avg %.4f ms  per-iter %.4f ms  alg GB/s %.1f" % (k, v["launches"] / steps, ms, per_it, gbs))
print("  sum per-iter %.4f ms" % tot, s.state())
