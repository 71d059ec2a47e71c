"""Runs the real-data examples shipped as golden fixtures with the reference's CLI parameters and prints one summary
line per problem (for profiles/*_real_data.log; reference numbers are the ones printed in its shipped logs)."""
import os, sys, time, gzip, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cuadmm_amd
from tests.conftest import load_npz_problem, GOLDEN
from tests.helpers import problem_to_amd
from oracle import cuadmm_oracle as orc          # TXT reader only (tools/, not the product path)

REF = {  # name: (mode switch_admm, max_iter, reference iterations, reference total seconds, reference ms/iter)
    "PlanarHand_N=1_MOMENT": (0, 20000, 878, 54.2, 56.3),
    "pendulum_N=80": (11000, 2000, None, None, 22.2),
    "1dc.1024": (11000, 20000, 353, 22.1, 62.4),
    "bqp-r1-40-1": (11000, 20000, 10397, 706.1, 67.9),
    "swissroll": (11000, 2000, None, None, 19.7),
    "ros_2000": (0, 20000, 3268, 3.4, 1.0),
    "PushT_N=10_MOMENT": (0, 3000, None, None, None),
}
def load(name):
    if os.path.exists(os.path.join(GOLDEN, "problems", name + ".npz")):
        return load_npz_problem(name)
    d = os.path.join(GOLDEN, "problems", name); t = tempfile.mkdtemp()
    for fn in os.listdir(d):
        with gzip.open(os.path.join(d, fn), "rb") as f, open(os.path.join(t, fn[:-3]), "wb") as g: shutil.copyfileobj(f, g)
    return orc.load_problem_txt(t)
for name, (sw, max_iter, rit, rsec, rms) in REF.items():
    p = load(name)
    s = cuadmm_amd.SDPSolver(verbose=False, profile=1)
    t = time.time(); s.init_problem(problem_to_amd(p)); ti = time.time() - t
    t = time.time(); s.solve(max_iter, 1e-3, 0, 50, 100, sw, 1.05); ts = time.time() - t
    it = s.info_iter_num; st = s.state(); pr = s.profile()
    parts = " ".join("%s %.3f" % (k, v["ms"] / it) for k, v in pr.items() if v["launches"])
    print("%-22s blocks %5d m %6d | %6d it%s  init %.2fs solve %.2fs  %.3f ms/it | ref: %s it, %s s, %s ms/it | maxfeas/relgap %.1e %.1e | per-iter ms: %s"
          % (name, len(p.blk), p.con_num, it, " (cap)" if it >= max_iter else "", ti, ts, ts / it * 1e3, rit, rsec, rms,
             max(st["errRp"], st["errRd"]), st["relgap"], parts), flush=True)
