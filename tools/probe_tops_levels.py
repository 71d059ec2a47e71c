"""Developer aid: dense tree tops at many cut heights and tail sizes against the plain sweeps / the host solve at the same tail
(30 iterations of every fixture given; the iterates must agree to the rounding of the differently associated sums)."""
import sys, os, tempfile, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cuadmm_amd
from tests import test_gpu_moment_parity as T
from tests.helpers import problem_to_amd

CASES = {"PlanarHand_N=1_MOMENT": [17152, 10240, 6144], "pendulum_N=80": [10496, 6144], "PushT_N=10_MOMENT": [14080, 8192],
         "PushBox_N=30_MOMENT": [18688, 8192, 4096]}
LEVELS = [2, 4, 8, 12, 24, 40]
names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(CASES)
worst = 0.0
for name in names:
    with tempfile.TemporaryDirectory() as td:
        p = T.load_problem(name, pathlib.Path(td))
    for k in CASES[name]:
        ref = None
        for L in [0] + LEVELS:
            s = cuadmm_amd.SDPSolver(verbose=False, options={"tail_k": k, "lead_tops": L})
            s.init_problem(problem_to_amd(p))
            s.solve(30, 0.0, 0, 50, 100, 15, 1.05)
            cur = {nm: s.info_arr(nm).copy() for nm in T.SIX}
            cur["X"] = s.X.copy()
            mode = int(s.counters()["dev_solve"])
            if L == 0:
                ref = cur
                print("%-24s tail %6d  reference: lead_tops = 0 (dev_solve %d)" % (name, k, mode), flush=True)
                continue
            dev = max(T.rel_dev(cur[nm], ref[nm], nm) for nm in T.SIX)
            dx = float(np.linalg.norm(cur["X"] - ref["X"]) / (1 + np.linalg.norm(ref["X"])))
            worst = max(worst, dev, dx)
            print("%-24s tail %6d  level %2d (dev_solve %d): max rel deviation %.1e, X %.1e" % (name, k, L, mode, dev, dx), flush=True)
print("worst", worst)
assert worst <= 1e-7
