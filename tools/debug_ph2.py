import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cuadmm_amd
from tests.conftest import load_npz_problem
from tests.helpers import problem_to_amd
p = load_npz_problem("PlanarHand_N=1_MOMENT")
s = cuadmm_amd.SDPSolver(verbose=True)
s.init_problem(problem_to_amd(p))
try:
    s.solve(int(sys.argv[1]) if len(sys.argv) > 1 else 100, 0.0, 0, 50, 100, 0, 1.05)
except cuadmm_amd.CuadmmError as e:
    print("ERR", e)
print(s.state())
