#!/usr/bin/env python3
"""Oracle trajectories of the MOMENT-RELAXATION inputs (run in the dev container; CPU only, minutes).

    python tests/golden/make_traj_moment.py [name ...]

Writes tests/golden/oracle_traj_moment.json: per-iteration (errRp, errRd, pobj, dobj, relgap, sig) of
oracle/cuadmm_oracle.py -- the exact sparse solve (SuperLU) and LAPACK dsyevd restating
src/solver.cu:478-500,534-647,693-729 -- in repr() precision for the first `iters` iterations, plus the same six
numbers at one LATE checkpoint and the norms of X, y, S there.  These are the inputs on which the engine's GPU
tail (tail_solve.hip), device lead solve (lead_solve.hip), long-row A^T y path and un-fused iteration run together;
tests/test_gpu_moment_parity.py compares the default engine against these numbers at a stated tolerance.

The inputs are the fixtures make_golden.py produced (tests/golden/problems/*), so nothing here reads /root/reference.
"""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import cuadmm_oracle as orc   # noqa: E402
from tests.conftest import load_npz_problem   # noqa: E402

COMMON = dict(sig=1.0, stop_tol=0.0, sig_update_threshold=0, sig_update_stage_1=50, sig_update_stage_2=100, sigscale=1.05)

# name -> (problem, switch_admm, head iterations, late checkpoint)
RUNS = {
    "PlanarHand_N=1_MOMENT/switch=0": ("PlanarHand_N=1_MOMENT", 0, 60, 300),
    "PlanarHand_N=1_MOMENT/switch=11000": ("PlanarHand_N=1_MOMENT", 11000, 60, 300),
    "pendulum_N=80/switch=11000": ("pendulum_N=80", 11000, 60, 1000),
    "pendulum_N=80/switch=11000/late=5000": ("pendulum_N=80", 11000, 60, 5000),      # round 4: the long row of the reference's log, backed further out
    "pendulum_N=80/switch=11000/late=20000": ("pendulum_N=80", 11000, 60, 20000),    # round 5: through the sGS -> ADMM switch at 11 000 and 9 000 iterations beyond
    "pendulum_N=80/switch=11000/late=100000": ("pendulum_N=80", 11000, 60, 100000),  # round 5: the whole run of examples/pendulum/N=80_licols.log (2 h on 8 cores)
    "PushT_N=10_MOMENT/switch=0": ("PushT_N=10_MOMENT", 0, 60, 500),
    "PushT_N=10_MOMENT/switch=11000": ("PushT_N=10_MOMENT", 11000, 60, 500),
    "PushT_N=30_MOMENT/switch=11000": ("PushT_N=30_MOMENT", 11000, 60, 200),
    # 14 blocks of 56 / 126 / 252, m = 3 002: the merged one-launch sign-path groups and the whole factor as the dense GPU tail
    "taha1a/switch=0": ("taha1a", 0, 60, 400),
    "taha1a/switch=11000": ("taha1a", 11000, 60, 400),
    # round 4: the reference's largest shipped moment relaxations (examples/SPOT/data/MOSEK/*.mat through make_large_moment.py)
    "PushBox_N=30_MOMENT/switch=11000": ("PushBox_N=30_MOMENT", 11000, 60, 200),
    "PushBox_N=50_MOMENT/switch=11000": ("PushBox_N=50_MOMENT", 11000, 60, 200),
    "PlanarHand_N=10_MOMENT/switch=11000": ("PlanarHand_N=10_MOMENT", 11000, 40, 60),
}


def load(name):
    d = os.path.join(HERE, "problems", name)
    if os.path.isdir(d):
        import gzip
        import shutil
        import tempfile
        tmp = tempfile.mkdtemp()
        for fn in os.listdir(d):
            with gzip.open(os.path.join(d, fn), "rb") as f, open(os.path.join(tmp, fn[:-3]), "wb") as g:
                shutil.copyfileobj(f, g)
        return orc.load_problem_txt(tmp + "/")
    return load_npz_problem(name)


def host_ldlt_solver(At_csr, relres_log):
    """y = (A A^T + 1e-15 I)^-1 rhs through the library's host LDL^T (CPU only).  The product code is only the SOLVER here: every y it
    returns is checked with scipy alone -- || (A A^T + 1e-15 I) y - rhs || / || rhs || by two sparse matrix-vector products, appended to
    `relres_log` -- so the trajectory does not rest on the library being right (tests/test_oracle_pinning.py asserts <= 1e-10 on what
    was recorded; contract: include/cuadmm/cholesky_cpu.h:146-155)."""
    import ctypes as C
    import cuadmm_amd
    from cuadmm_amd._lib import check
    lib = cuadmm_amd.load()
    L, m = At_csr.shape
    rp = np.ascontiguousarray(At_csr.indptr, np.int32)
    ci = np.ascontiguousarray(At_csr.indices, np.int32)
    v = np.ascontiguousarray(At_csr.data, np.float64)
    h = C.c_void_p()
    check(lib.cuadmm_aat_create(int(m), int(L), rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), 1e-15, C.byref(h)))
    perm = np.ctypeslib.as_array(lib.cuadmm_aat_perm(h), shape=(m,)).copy()
    A_csr = At_csr.T.tocsr()                             # scipy's own copy: the check shares no code with the factor

    def solve(rhs):
        rhs = np.asarray(rhs, np.float64)
        r = np.ascontiguousarray(rhs[perm])
        out = np.empty(m)
        check(lib.cuadmm_aat_solve_permuted(h, r.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
        y = np.empty(m)
        y[perm] = out
        nr = float(np.linalg.norm(rhs))
        relres_log.append(float(np.linalg.norm(A_csr @ (At_csr @ y) + 1e-15 * y - rhs)) / nr if nr > 0 else 0.0)
        return y
    solve._keep = (h, rp, ci, v)
    return solve


def main():
    out_path = os.path.join(HERE, "oracle_traj_moment.json")
    traj = json.load(open(out_path)) if os.path.exists(out_path) else {}
    want = sys.argv[1:] or list(RUNS)
    for key in want:
        prob, sw, head, late = RUNS[key]
        p = load(prob)
        host_factor = prob in ("PushT_N=30_MOMENT", "PlanarHand_N=10_MOMENT")      # factors SuperLU cannot hold
        if host_factor:
            # SuperLU runs out of memory on this A A^T (m = 53 290, also in symmetric mode with minimum degree; the reference's
            # CHOLMOD needed 369 s).  The exact sparse solve of this one trajectory is the library's HOST factor instead (CPU code:
            # cuadmm_aat_create / cuadmm_aat_solve_permuted, the CHOLMOD contract of cholesky_cpu.h:62-155; no GPU, no explicit
            # inverse) -- everything else is the numpy / LAPACK oracle.  The engine's default path for this input uses the GPU
            # tail (explicit inverse) + device sweeps, so the comparison still crosses two different solvers.
            import scipy.sparse.linalg as spla
            _real_factorized = spla.factorized
            orc.spla.factorized = lambda M: None
        t0 = time.time()
        s = orc.OracleSolver().init_problem(p)
        if host_factor:
            orc.spla.factorized = _real_factorized
            relres = []
            s._solve = host_ldlt_solver(s.At_csr, relres)
        t1 = time.time()
        info = s.solve(late, 0.0, 0, 50, 100, sw, 1.05)
        t2 = time.time()
        six = ("errRp", "errRd", "pobj", "dobj", "relgap", "sig")
        rec = dict(problem=prob, iters=head, late=late, params=dict(COMMON, switch_admm=sw),
                   init=dict(norm_borg=repr(s.norm_borg), norm_Corg=repr(s.norm_Corg), bscale=repr(s.bscale), Cscale=repr(s.Cscale)))
        for nm in six:
            arr = getattr(info, nm)
            rec[nm] = [repr(float(x)) for x in arr[:head]]
            rec["late_" + nm] = repr(float(arr[late - 1]))
        # the late checkpoint's iterates (unscaled, as SDPSolver::solve leaves them)
        rec["late_X_norm"] = repr(float(np.linalg.norm(s.X)))
        rec["late_y_norm"] = repr(float(np.linalg.norm(s.y)))
        rec["late_S_norm"] = repr(float(np.linalg.norm(s.S)))
        if host_factor:
            # one entry per y-solve, in call order (two per sGS iteration), scipy-only residual of the library's host factor
            rec["ysolve_solver"] = "cuadmm_aat_create / cuadmm_aat_solve_permuted (host LDL^T); verified per solve by scipy matvecs"
            rec["ysolve_count"] = len(relres)
            rec["ysolve_relres_max"] = repr(max(relres))
            rec["ysolve_relres_head"] = ["%.3e" % x for x in relres[:2 * head]]
        traj[key] = rec
        print("%s: init %.1fs, %d iterations %.1fs" % (key, t1 - t0, late, t2 - t1), flush=True)
        with open(out_path, "w") as f:
            json.dump(traj, f, indent=1)


if __name__ == "__main__":
    main()
