"""Round 6: what do C4's small blocks cost?  BASELINE configs[3] whole, without its n <= 15 blocks, and those blocks alone --
ADMM iterations per second of each (closed-block kernels, one launch per iteration and size class, classes on concurrent streams)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cuadmm_amd
from cuadmm_amd import synthetic

blk = synthetic.config_c4_blk(100000)
lib = cuadmm_amd.load()
for name, sel in (("all six sizes", blk > 0), ("n = 28, 45 only", blk >= 28), ("n = 45 only", blk == 45), ("n <= 15 only", blk <= 15), ("n = 3, 6 only", blk <= 6), ("n = 10, 15 only", (blk >= 10) & (blk <= 15))):
    b = np.ascontiguousarray(blk[sel])
    p = synthetic.make_synthetic(b, cons_per_block=3, seed=20240601)
    s = cuadmm_amd.SDPSolver(verbose=False, profile=2)
    s.init_problem(cuadmm_amd.Problem(p.vec_len, p.con_num, p.blk, p.At_col_ptrs, p.At_row_ids, p.At_vals, p.b_idx, p.b_vals, p.C_idx, p.C_vals))
    s.solve(10, 0.0, 0, 50, 100, 0, 1.05)
    lib.cuadmm_dev_sync()
    t0 = time.perf_counter()
    s.solve(60, 0.0, 0, 50, 100, 0, 1.05, if_first=False)
    lib.cuadmm_dev_sync()
    dt = (time.perf_counter() - t0) / 60
    print("%-18s %6d blocks  %.3f ms per iteration  %.1f iters/s   closed %d" % (name, b.size, dt * 1e3, 1 / dt, s.counters()["closed_blocks"]), flush=True)
    del s
