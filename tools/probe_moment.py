"""Developer aid: moment-relaxation trajectories against the oracle under option variants."""
import sys, os, json, tempfile, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cuadmm_amd
from tests import test_gpu_moment_parity as T
from tests.helpers import problem_to_amd

keys = sys.argv[1].split(",") if len(sys.argv) > 1 else sorted(T.TRAJ)
variants = [{}, {"psd_hint": 0}, {"tail_k": 0}, {"host_solve": 1}, {"tiny_sign": 0}, {"psd_n16": 0, "psd_n32": 0}]
if len(sys.argv) > 2:          # variants on the command line: "tail_k=8192,lead_tops=32;lead_tops=0"
    variants = [dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in v.split(",") if kv) for v in sys.argv[2].split(";")]
for key in keys:
    rec = T.TRAJ[key]
    with tempfile.TemporaryDirectory() as td:
        p = T.load_problem(rec["problem"], pathlib.Path(td))
    for opt in variants:
        s = cuadmm_amd.SDPSolver(verbose=False, options=opt)
        s.init_problem(problem_to_amd(p))
        prm = rec["params"]
        s.solve(int(rec["late"]), 0.0, prm["sig_update_threshold"], prm["sig_update_stage_1"], prm["sig_update_stage_2"],
                prm["switch_admm"], prm["sigscale"])
        h, l, ok = T.deviations(s, rec)
        c = s.counters()
        print(key, opt, "tail_k", c["tail_k"], "dev", c["dev_solve"], "inv_resid %.1e" % s.tail_info()["inverse_residual"], "head", {k: "%.1e" % v for k, v in h.items()},
              "late", {k: "%.1e" % v for k, v in l.items()}, ok, flush=True)
        got = s.info_arr("pobj"); ref = np.array([float(x) for x in rec["pobj"]])
        d = np.abs(got[:len(ref)] - ref)
        print("   pobj absdiff first 12:", " ".join("%.1e" % x for x in d[:12]), "max at", int(np.argmax(d)))
        del s
