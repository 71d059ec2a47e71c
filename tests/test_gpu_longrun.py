"""Long-run parity on real data ("thin long-run end of parity"): whole solves of the reference's shipped examples
to ITS stopping iteration, compared with the iteration counts and the last printed table row of its console logs
(tests/golden/ref_logs.json, transcribed by tests/golden/make_golden.py from examples/benchmarks/**/*.log,
examples/plato/logs/*.log, examples/pendulum/N=80_licols.log).

Bar: stopping iteration within max(2, 0.5 %) of the reference's, the last printed row to >= 3 significant digits (the
reference's own small-block eigensolver stops at tol 1e-6, include/cuadmm/cusolver.h:112-123, so its late digits carry that
noise), final objectives within the stopping tolerance.  Run times are printed (pytest -s) for profiles/r02_real_data.log.
"""
import os
import time

import numpy as np
import pytest

import cuadmm_amd
from oracle import cuadmm_oracle as orc
from tests.conftest import load_npz_problem
from tests.helpers import problem_to_amd

pytestmark = pytest.mark.gpu


def _load(name, problem_dirs):
    if name in problem_dirs:
        return orc.load_problem_txt(problem_dirs[name])
    return load_npz_problem(name)


def _solve(p, max_iter, stop_tol, switch_admm):
    s = cuadmm_amd.SDPSolver(verbose=False)
    t0 = time.time()
    s.init_problem(problem_to_amd(p))
    t1 = time.time()
    s.solve(max_iter, stop_tol, 0, 50, 100, switch_admm, 1.05)
    return s, t1 - t0, time.time() - t1


def _check_last_row(s, row, rel):
    it = s.info_iter_num
    got = [s.info_arr(n)[it - 1] for n in ("errRp", "errRd", "pobj", "dobj", "relgap", "sig")]
    want = [float(x) for x in (row[1], row[2], row[3], row[4], row[5], row[7])]
    for name, g, w, r in zip(("errRp", "errRd", "pobj", "dobj", "relgap", "sig"), got, want, (rel, rel, rel, rel, rel, 6e-2)):
        if name == "errRp" and w < 1e-9:
            # sGS phase: the reference's primal residual sits at the roundoff floor of its exact CHOLMOD solve (printed 8e-13);
            # with the dense tail of the factor applied as an explicit inverse on the GPU (tail_solve.hip) it sits at the
            # inverse's accuracy instead (1.9e-7 on PushT_N=30, whose Schur complement is numerically singular) -- four
            # orders below the stopping tolerance, same iteration count
            assert g < 1e-6, (name, g, w)
        else:
            # residuals below ~1e-5 carry the noise of the reference's own eigensolver tolerance (1e-6, cusolver.h:112-123)
            atol = 3e-7 if name in ("errRp", "errRd") else (2e-5 if name in ("pobj", "dobj", "relgap") else 1e-10)
            assert abs(g - w) <= r * abs(w) + atol, (name, g, w, got, want)


@pytest.mark.parametrize("key,rel", [
    ("PushT_N=10_MOMENT/cuADMM", 2e-2), ("PushT_N=10_MOMENT/sGS", 2e-2),      # 7 237 / 6 149 iterations
    ("PlanarHand_N=1_MOMENT/cuADMM", 2e-2), ("PlanarHand_N=1_MOMENT/sGS", 2e-2),   # 878 / 800
    ("ros_2000/cuADMM", 2e-2), ("ros_2000/sGS", 2e-2),                        # 3 268 / 12 732
    ("rose13/sGS", 2e-2),                                                     # 59 360
    ("chs_5000/cuADMM", 2e-2), ("chs_5000/sGS", 2e-2),                        # SeDuMi .mat through the converter
    ("PushT_N=30_MOMENT/sGS", 2e-2),                                          # MOSEK .mat; 369 s before iteration 0 in the reference
])
def test_runs_to_the_reference_stopping_iteration(key, rel, ref_logs, problem_dirs):
    lg = ref_logs[key]
    p = _load(lg["problem"], problem_dirs)
    ref_it = int(lg["rows"][-1][0])
    s, t_init, t_solve = _solve(p, 200000, lg["params"]["stop_tol"], lg["params"]["switch_admm"])
    it = s.info_iter_num
    print("\n[longrun] %-30s reference %6d it | here %6d it  init %.2f s  solve %.2f s (%.3f ms/it)  final maxfeas %.1e relgap %.1e"
          % (key, ref_it, it, t_init, t_solve, t_solve / max(it, 1) * 1e3, max(s.state()["errRp"], s.state()["errRd"]), s.state()["relgap"]))
    assert abs(it - ref_it) <= max(2, int(0.005 * ref_it)), (key, it, ref_it)
    if it == ref_it:
        _check_last_row(s, lg["rows"][-1], rel)
    tol = lg["params"]["stop_tol"]
    for name in ("pobj", "dobj"):
        w = float(lg["final"][name])
        assert abs(s.state()[name] - w) <= 2 * tol * (1 + abs(w)), (name, s.state()[name], w)


def test_taha1a_against_the_oracle_and_mosek():
    """examples/plato/MATLAB/taha1a.mat (SeDuMi; 14 blocks of 56 / 126 / 252, m = 3002) through the converter.  The reference ships
    no cuADMM log for it, only MOSEK's (examples/benchmarks/taha1a/MOSEK.log: optimum -1.0000000103 / -1.0000000154, kept in the
    fixture): an INDEPENDENT solver's answer.  The trajectory is pinned to the oracle's (1e-8 relative over the solve to 1e-3: blocks
    on the one-wavefront, the one-launch cluster and the batched-GEMM projection paths at once), the optimum of a solve to 1e-4 to
    MOSEK's within twice the stopping tolerance."""
    import os
    from tests.conftest import GOLDEN
    p = load_npz_problem("taha1a")
    d = np.load(os.path.join(GOLDEN, "problems", "taha1a.npz"))
    o = orc.OracleSolver().init_problem(p)
    o.solve(3000, 1e-3, 0, 50, 100, 11000, 1.05)
    s, _, _ = _solve(p, 3000, 1e-3, 11000)
    n = len(o.info.pobj)
    assert s.info_iter_num == n                                               # 137 iterations
    for nm in ("errRp", "errRd", "pobj", "dobj", "relgap"):
        got, ref = np.asarray(s.info_arr(nm))[:n], np.asarray(getattr(o.info, nm))
        err = np.abs(got - ref) / (1e-3 + np.abs(ref))
        assert np.max(err) <= 1e-8, (nm, float(np.max(err)), int(np.argmax(err)))
    tol = 1e-4
    s, t_init, t_solve = _solve(p, 20000, tol, 11000)
    it = s.info_iter_num
    print("\n[longrun] %-30s oracle     656 it | here %6d it  init %.2f s  solve %.2f s (%.3f ms/it)  pobj %.9f dobj %.9f (MOSEK %.10f / %.10f)"
          % ("taha1a/sGS tol 1e-4", it, t_init, t_solve, t_solve / max(it, 1) * 1e3, s.state()["pobj"], s.state()["dobj"], float(d["mosek_pobj"]), float(d["mosek_dobj"])))
    assert abs(it - 656) <= 7                                                 # the oracle's stopping iteration, 1 %
    for name, w in (("pobj", float(d["mosek_pobj"])), ("dobj", float(d["mosek_dobj"]))):
        assert abs(s.state()[name] - w) <= 2 * tol * (1 + abs(w)), (name, s.state()[name], w)
