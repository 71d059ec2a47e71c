"""The MATLAB front end behind its C entry (MEX contract): cuadmm_mex_call takes exactly what the reference's
mexFunction marshals out of its mxArrays (MATLAB/cuadmm_MATLAB.cu:197-293: size_t jc / ir, sparse b and C, blk as doubles,
dense X0 / y0 / S0) and returns what it packs into [X, y, S, info] (:366-424).  Checked against the oracle run with the
MEX's EFFECTIVE defaults: threshold 500, stages 50 / 100, switch_admm 11000, sigscale 1.0 (the optional arguments are read
only when nlhs >= 12..16, i.e. never, :297-333), X0 / y0 / S0 always handed to init."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

import cuadmm_amd
from cuadmm_amd._lib import check
from oracle import cuadmm_oracle as orc

lib = cuadmm_amd.load()
P = lambda a: a.ctypes.data_as(C.c_void_p)


def _marshal(p):
    """what MATLAB holds: At sparse (vec_len x con_num) with size_t jc / ir; b, C sparse column vectors; blk double"""
    At = sp.csc_matrix((p.At_vals, p.At_row_ids, p.At_col_ptrs), shape=(p.vec_len, p.con_num))
    At.sort_indices()
    b = sp.csc_matrix((p.b_vals, (p.b_idx, np.zeros(len(p.b_idx), int))), shape=(p.con_num, 1))
    Cv = sp.csc_matrix((p.C_vals, (p.C_idx, np.zeros(len(p.C_idx), int))), shape=(p.vec_len, 1))
    u64 = lambda a: np.ascontiguousarray(a, dtype=np.uint64)
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    return dict(At=(u64(At.indptr), u64(At.indices), f64(At.data)), b=(u64(b.indptr), u64(b.indices), f64(b.data)),
                C=(u64(Cv.indptr), u64(Cv.indices), f64(Cv.data)), blk=f64(p.blk))


def _mex(p, max_iter, stop_tol, X0, y0, S0, sig, nlhs=4, optional5=None, eig_streams=15):
    m = _marshal(p)
    res = C.c_void_p()
    opt = np.ascontiguousarray(optional5, dtype=np.float64) if optional5 is not None else None
    rc = lib.cuadmm_mex_call(eig_streams, max_iter, stop_tol,
                             p.vec_len, p.con_num, P(m["At"][0]), P(m["At"][1]), P(m["At"][2]),
                             p.con_num, P(m["b"][0]), P(m["b"][1]), P(m["b"][2]),
                             p.vec_len, P(m["C"][0]), P(m["C"][1]), P(m["C"][2]),
                             len(p.blk), P(m["blk"]),
                             X0.size, P(X0), y0.size, P(y0), S0.size, P(S0),
                             sig, nlhs, P(opt) if opt is not None else None, C.byref(res))
    if rc != 0:
        return rc, None
    vl, cn, it, tt = C.c_int(), C.c_int(), C.c_int(), C.c_double()
    check(lib.cuadmm_mex_result_dims(res, C.byref(vl), C.byref(cn), C.byref(it), C.byref(tt)))
    out = dict(X=np.empty(vl.value), y=np.empty(cn.value), S=np.empty(vl.value), iter_num=it.value, total_time=tt.value)
    check(lib.cuadmm_mex_result_XyS(res, P(out["X"]), P(out["y"]), P(out["S"])))
    for w, name in enumerate(["pobj", "dobj", "errRp", "errRd", "relgap", "sig", "bscale", "Cscale"]):
        a = np.empty(it.value)
        assert lib.cuadmm_mex_result_info(res, w, P(a)) == it.value
        out[name] = a
    lib.cuadmm_mex_result_free(res)
    return 0, out


def test_mex_entry_rejects_inconsistent_sizes_without_touching_a_device(problem_dirs):
    p = orc.load_problem_txt(problem_dirs["hinf12"])
    z = np.zeros(p.vec_len)
    rc, _ = _mex(p, 5, 1e-3, z[:-1].copy(), np.zeros(p.con_num), z, 1.0)     # X0 one entry short
    assert rc == -1 and b"X0" in lib.cuadmm_last_error()


@pytest.mark.gpu
def test_mex_call_matches_the_oracle_with_the_mex_effective_defaults(problem_dirs):
    p = orc.load_problem_txt(problem_dirs["hinf12"])
    rng = np.random.default_rng(0)
    # a warm start as a MATLAB caller would pass it: the iterate of a short previous solve
    o0 = orc.OracleSolver().init_problem(p)
    o0.solve(30, 0.0, 500, 50, 100, 11000, 1.0)
    X0, y0, S0, sig0 = o0.X.copy(), o0.y.copy(), o0.S.copy(), 1.3
    # the caller passes sigscale = 1.05 and switch_admm = 20: both must be IGNORED (nlhs = 4 < 12)
    rc, r = _mex(p, 120, 0.0, X0, y0, S0, sig0, nlhs=4, optional5=[7, 3, 5, 20, 1.05])
    assert rc == 0
    o = orc.OracleSolver().init_problem(p, X=X0, y=y0, S=S0, sig=sig0)
    info = o.solve(120, 0.0, 500, 50, 100, 11000, 1.0)
    assert r["iter_num"] == 120 and r["total_time"] > 0
    for name, ref in (("pobj", info.pobj), ("dobj", info.dobj), ("errRp", info.errRp), ("errRd", info.errRd), ("relgap", info.relgap)):
        ref = np.asarray(ref)
        assert np.max(np.abs(r[name] - ref) / (1e-9 + np.abs(ref))) <= 1e-7, name
    assert np.array_equal(r["sig"], np.asarray(info.sig))               # sigscale 1.0: sigma never moves
    assert np.all(r["sig"] == sig0)
    assert np.max(np.abs(r["X"] - o.X)) <= 1e-8 * (1 + np.max(np.abs(o.X)))
    assert np.max(np.abs(r["y"] - o.y)) <= 1e-7 * (1 + np.max(np.abs(o.y)))
    assert np.max(np.abs(r["S"] - o.S)) <= 1e-8 * (1 + np.max(np.abs(o.S)))
    # with the reference's conditions met (nlhs >= 16) the optional values WOULD be used: sigma then moves
    rc, r2 = _mex(p, 120, 0.0, X0, y0, S0, sig0, nlhs=16, optional5=[0, 10, 10, 11000, 1.05])
    assert rc == 0 and not np.all(r2["sig"] == sig0)
