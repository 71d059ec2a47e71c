"""The reference's largest shipped problems on the engine with the reference CLI's parameters (main.cu:23,39: sig = 1, stop_tol = 1e-3,
sig_update 0 / 50 / 100, switch_admm as given): init broken out, iterations and time to the tolerance, per-phase milliseconds.
    python tools/run_large.py <fixture> [switch_admm] [max_iter] [key=value ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cuadmm_amd
from tests.conftest import load_npz_problem
from tests.helpers import problem_to_amd
name = sys.argv[1]
sw = int(sys.argv[2]) if len(sys.argv) > 2 else 11000
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 30000
opts = {}
for kv in sys.argv[4:]:
    k, v = kv.split("="); opts[k] = float(v)
if os.path.isdir(os.path.join(ROOT, "tests", "golden", "problems", name)):      # a TXT fixture (blk.txt.gz ...)
    import pathlib, tempfile
    from tests.test_gpu_moment_parity import load_problem
    p = load_problem(name, pathlib.Path(tempfile.mkdtemp()))
else:
    p = load_npz_problem(name)
s = cuadmm_amd.SDPSolver(verbose=False, profile=1, options=opts)
t = time.time(); s.init_problem(problem_to_amd(p)); ti = time.time() - t
t = time.time(); s.solve(cap, 1e-3, 0, 50, 100, sw, 1.05); ts = time.time() - t
st = s.state()
it = s.info_iter_num
prof = {k: round(v["ms"] / max(it, 1), 4) for k, v in s.profile().items() if v["launches"]}
print("RESULT %s switch_admm=%d %s: L %d m %d | init %.2f s | %d iterations in %.2f s (%.3f ms/iter) | errRp %.2e errRd %.2e relgap %.2e pobj %.9e dobj %.9e | tail_k %d dev_solve %d | %s"
      % (name, sw, opts, p.vec_len, p.con_num, ti, it, ts, ts / max(it, 1) * 1e3, st["errRp"], st["errRd"], st["relgap"], st["pobj"], st["dobj"], int(s.counters()["tail_k"]), int(s.counters()["dev_solve"]), prof), flush=True)
