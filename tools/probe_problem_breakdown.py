"""Per-phase milliseconds per iteration of a fixture problem (tests/golden/problems/<name>.npz or a TXT directory fixture).
python tools/probe_problem_breakdown.py taha1a [iterations]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import cuadmm_amd  # noqa: E402
from tests.conftest import load_npz_problem  # noqa: E402
from tests.helpers import problem_to_amd  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "taha1a"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
mode = sys.argv[3] if len(sys.argv) > 3 else "sgs"           # sgs | admm
from tests.conftest import GOLDEN  # noqa: E402
if os.path.isdir(os.path.join(GOLDEN, "problems", name)):    # a TXT directory fixture (*.txt.gz)
    import gzip
    import shutil
    import tempfile
    from oracle import cuadmm_oracle as orc
    tmp = tempfile.mkdtemp()
    for fn in os.listdir(os.path.join(GOLDEN, "problems", name)):
        with gzip.open(os.path.join(GOLDEN, "problems", name, fn), "rb") as fi, open(os.path.join(tmp, fn[:-3]), "wb") as fo:
            shutil.copyfileobj(fi, fo)
    p = orc.load_problem_txt(tmp + "/")
else:
    p = load_npz_problem(name)
s = cuadmm_amd.SDPSolver(verbose=False, options={"profile": 1})
s.init_problem(problem_to_amd(p))
sw = 11000 if mode == 'sgs' else 0
s.solve(50, 0.0, 0, 50, 100, sw, 1.05)
s.reset_profile()
t0 = time.perf_counter()
s.solve(iters, 0.0, 0, 50, 100, sw, 1.05, if_first=False)
dt = time.perf_counter() - t0
print("%s: %.3f ms per iteration (%s), m = %d, %d blocks (max n %d), plan %s" % (name, dt / iters * 1e3, mode, p.con_num, len(p.blk), int(max(p.blk)), {k: v for k, v in s.counters().items() if v}))
for k, v in s.profile().items():
    if v["launches"]:
        print("  %-16s %8.0f launches  %8.4f ms per iteration" % (k, v["launches"], v["ms"] / iters))
