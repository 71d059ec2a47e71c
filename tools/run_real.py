"""Runs the engine on the real-data configurations (PlanarHand_N=1 = BASELINE config 1 data, pendulum N=80 = config 5)
with the reference CLI parameters and prints iterations/time (for BASELINE.md)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cuadmm_amd
from tests.conftest import load_npz_problem
from tests.helpers import problem_to_amd
name = sys.argv[1]; max_iter = int(sys.argv[2]); sw = int(sys.argv[3]); tol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-3
p = load_npz_problem(name)
s = cuadmm_amd.SDPSolver(verbose=True, profile=1)
t = time.time(); s.init_problem(problem_to_amd(p)); ti = time.time() - t
t = time.time(); s.solve(max_iter, tol, 0, 50, 100, sw, 1.05); ts = time.time() - t
print("RESULT %s init %.2fs solve %.2fs iters %d -> %.2f ms/iter" % (name, ti, ts, s.info_iter_num, ts / max(s.info_iter_num, 1) * 1e3))
for k, v in s.profile().items():
    if v["launches"]: print("   %-12s per-iter %.3f ms" % (k, v["ms"] / s.info_iter_num))
print(s.state())
