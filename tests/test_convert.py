"""Input-side converters (cuadmm_amd/convert.py) against the inputs the reference ships in BOTH forms: the source format
(SeDuMi .mat, MOSEK struct .mat, svec-form .mat, SDPA .dat-s; tests/golden/formats) and the TXT directory its MATLAB
converters produced from it (tests/golden/problems).  No GPU needed."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import cuadmm_amd
from cuadmm_amd import convert
from oracle import cuadmm_oracle as orc
from tests.conftest import GOLDEN

FMT = os.path.join(GOLDEN, "formats")


def _dense(p):
    if hasattr(p, "At_csc_vals"):
        A = sp.csc_matrix((p.At_csc_vals, p.At_csc_row_ids, p.At_csc_col_ptrs), shape=(p.vec_len, p.con_num))
        b = np.zeros(p.con_num); b[p.b_indices] = p.b_vals
        c = np.zeros(p.vec_len); c[p.C_indices] = p.C_vals
        return np.asarray(p.blk_vals), A, b, c
    A = sp.csc_matrix((p.At_vals, p.At_row_ids, p.At_col_ptrs), shape=(p.vec_len, p.con_num))
    b = np.zeros(p.con_num); b[p.b_idx] = p.b_vals
    c = np.zeros(p.vec_len); c[p.C_idx] = p.C_vals
    return np.asarray(p.blk), A, b, c


def _same(p, q, tol):
    bp, Ap, b1, c1 = _dense(p)
    bq, Aq, b2, c2 = _dense(q)
    assert np.array_equal(bp, bq) and Ap.shape == Aq.shape
    assert Ap.nnz == Aq.nnz and np.array_equal(Ap.indptr, Aq.indptr) and np.array_equal(Ap.indices, Aq.indices)   # svec slots bit-exact
    assert np.max(np.abs(Ap.data - Aq.data)) <= tol
    assert np.max(np.abs(b1 - b2)) <= tol and np.max(np.abs(c1 - c2)) <= tol


@pytest.mark.parametrize("name,loader,tol", [
    ("hinf12", lambda: convert.problem_from_sedumi_mat(os.path.join(FMT, "hinf12_sedumi.mat")), 1e-14),
    ("truss5", lambda: convert.problem_from_sedumi_mat(os.path.join(FMT, "truss5_sedumi.mat")), 1e-15),
    ("PushT_N=10_MOMENT", lambda: convert.problem_from_mosek_mat(os.path.join(FMT, "PushT_N=10_MOMENT_mosek.mat")), 1e-15),
    ("biggs", lambda: convert.problem_from_sdpa(os.path.join(FMT, "biggs.dat-s.gz")), 1e-13),
])
def test_converters_reproduce_the_shipped_txt_inputs(name, loader, tol, problem_dirs):
    _same(loader(), orc.load_problem_txt(problem_dirs[name]), tol)


def test_svec_form_mat_and_txt_round_trip(problem_dirs, tmp_path):
    blk = [n for _, n in orc.read_blk(problem_dirs["rose13"] + "blk.txt")]
    p = convert.problem_from_svec_mat(os.path.join(FMT, "rose13_svec.mat"), blk)
    _same(p, orc.load_problem_txt(problem_dirs["rose13"]), 0.0)
    # write_txt -> the engine's own TXT loader (io.cpp) -> identical arrays
    convert.write_txt(p, str(tmp_path / "out"))
    q = cuadmm_amd.Problem.from_txt(str(tmp_path / "out") + "/")
    _same(p, q, 0.0)


def test_sedumi_vec_to_svec_on_a_hand_example():
    # one 2x2 block, X = [x11 x12; x21 x22] as vec (x11, x21, x12, x22); constraint <[[1, 2], [4, 5]], X> = 3
    A = np.array([[1.0, 4.0, 2.0, 5.0]])
    p = convert.problem_from_sedumi(A, np.array([3.0]), np.array([1.0, 0.0, 0.0, 1.0]), {"s": [2]})
    _, At, b, c = _dense(p)
    # svec order (1,1), (1,2), (2,2); off-diagonal: sqrt2 * (2 + 4) / 2
    assert np.allclose(At.toarray().ravel(), [1.0, convert.SQRT2 * 3.0, 5.0], rtol=0, atol=1e-15)
    assert np.array_equal(b, [3.0]) and np.array_equal(c, [1.0, 0.0, 1.0])


def test_unsupported_cones_are_rejected():
    A = np.ones((1, 5))
    with pytest.raises(ValueError, match="K.l"):
        convert.problem_from_sedumi(A, np.ones(1), np.ones(5), {"l": [1], "s": [2]})
    with pytest.raises(ValueError, match="K.s"):
        convert.problem_from_sedumi(A, np.ones(1), np.ones(5), {"s": []})


def test_sdpa_diagonal_blocks_are_rejected(tmp_path):
    f = tmp_path / "diag.dat-s"
    f.write_text("1\n2\n2 -3\n1.0\n0 1 1 1 1.0\n1 1 1 1 1.0\n")
    with pytest.raises(ValueError, match="diagonal"):
        convert.problem_from_sdpa(str(f))
