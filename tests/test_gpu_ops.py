"""GPU parity of the op-level entry points, one per reference kernel / library wrapper,
against the oracle and the reference's own unit-test vectors.  Integer/index work is bit-exact."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

import cuadmm_amd
from cuadmm_amd._lib import check
from oracle import cuadmm_oracle as orc
from tests.conftest import GOLDEN
from tests.helpers import Dev

pytestmark = pytest.mark.gpu
lib = cuadmm_amd.load()


def test_matrices_to_vector_reference_kat():
    # test/kernels_test.hpp:224-308
    mom = np.array([1, 2, 3, 4, 2, 5, 6, 7, 3, 6, 8, 9, 4, 7, 9, 10], float)
    loc = np.array([2, 3, 4, 5, 3, 6, 7, 8, 4, 7, 9, 10, 5, 8, 10, 11], float)
    m1 = [0, 1, 2, 3, 5, 6, 7, 10, 11, 15]; m2 = [0, 4, 8, 12, 5, 9, 13, 10, 14, 15]
    mB = np.array([0] * 10 + [1] * 10, np.int32); M1 = np.array(m1 + m1, np.int32); M2 = np.array(m2 + m2, np.int32)
    dX = Dev(shape=(20,)); dm, dl = Dev(mom), Dev(loc)
    dB, d1, d2 = Dev(mB), Dev(M1), Dev(M2)
    check(lib.cuadmm_op_matrices_to_vector(dX.ptr, dm.ptr, dl.ptr, dB.ptr, d1.ptr, d2.ptr, 20, None))
    s = orc.SQRT2
    want = [1, 2 * s, 3 * s, 4 * s, 5, 6 * s, 7 * s, 8, 9 * s, 10, 2, 3 * s, 4 * s, 5 * s, 6, 7 * s, 8 * s, 9, 10 * s, 11]
    assert dX.get().tolist() == want                                   # EXPECT_EQ in the reference: bit exact


@pytest.mark.parametrize("blk", [[2, 4], [1, 2, 3, 4], [6] * 50 + [4], [7, 10, 28, 55, 3, 28, 120]])
def test_vec_mat_roundtrip_with_reference_maps(blk):
    """vector_to_matrices -> matrices_to_vector with the maps of get_maps (bit-exact vs the oracle)."""
    blk = np.array(blk, np.int32)
    L = int(orc.svec_block_offsets(blk)[-1])
    mB, m1, m2 = (np.zeros(L, np.int32) for _ in range(3))
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    check(lib.cuadmm_get_maps(P(blk), blk.size, L, P(mB), P(m1), P(m2)))
    sizes, nums = orc.analyze_blk(blk)
    ms = orc.MatrixSizes(sizes, nums)
    x = np.random.default_rng(0).standard_normal(L)
    lg, smm = np.zeros(max(ms.total_large_mat_size, 1)), np.zeros(max(ms.total_small_mat_size, 1))
    orc.vector_to_matrices(x, lg, smm, mB, m1, m2)
    dx, dlg, dsm = Dev(x), Dev(np.zeros_like(lg)), Dev(np.zeros_like(smm))
    dB, d1, d2 = Dev(mB), Dev(m1), Dev(m2)
    check(lib.cuadmm_op_vector_to_matrices(dx.ptr, dlg.ptr, dsm.ptr, dB.ptr, d1.ptr, d2.ptr, L, None))
    assert np.array_equal(dlg.get(), lg) and np.array_equal(dsm.get(), smm)
    dback = Dev(shape=(L,))
    check(lib.cuadmm_op_matrices_to_vector(dback.ptr, dlg.ptr, dsm.ptr, dB.ptr, d1.ptr, d2.ptr, L, None))
    assert np.array_equal(dback.get(), orc.matrices_to_vector(lg, smm, mB, m1, m2))
    assert np.max(np.abs(dback.get() - x)) <= 4e-16 * np.max(np.abs(x))     # sqrt2*sqrt2inv round trip


def test_permutation_scatter_kat():
    # test/kernels_test.hpp:4-33: v1[perm[i]] = v2[i]
    v2 = np.arange(10, dtype=float); perm = np.array([6, 4, 1, 3, 0, 5, 2, 8, 7, 9], np.int32)
    d1, d2, dp = Dev(np.zeros(10)), Dev(v2), Dev(perm)
    check(lib.cuadmm_op_permute(d1.ptr, d2.ptr, dp.ptr, 10, None))
    out = d1.get()
    assert all(out[perm[i]] == v2[i] for i in range(10))


def test_normA_kat():
    # test/kernels_test.hpp:35-83 with test/data/sparse_matrix_coo.txt
    r, c, v = orc.read_coo(os.path.join(GOLDEN, "io", "sparse_matrix_coo.txt"))
    cp, ri, vv = orc.coo_to_csc(c, r, v, 4)
    dcp, dv, dn = Dev(cp), Dev(vv), Dev(shape=(4,))
    check(lib.cuadmm_op_get_normA(dcp.ptr, dv.ptr, dn.ptr, 4, None))
    assert dn.get().tolist() == [np.sqrt(1000.0), np.sqrt(4000.0), 40.0, 50.0]
    assert np.allclose(dv.get(), [10 / np.sqrt(1000), 30 / np.sqrt(1000), 20 / np.sqrt(4000), 60 / np.sqrt(4000), 1, 1], rtol=1e-15)


def test_max_zero_and_mul_diag_batch():
    rng = np.random.default_rng(1)
    w = rng.standard_normal(1000)
    dw = Dev(w)
    check(lib.cuadmm_op_max_zero(dw.ptr, 1000, None))
    assert np.array_equal(dw.get(), np.maximum(w, 0.0))
    n, cnt = 5, 7                                                        # diagonal_batch.cu:11-23: scales COLUMN j
    V = rng.standard_normal((cnt, n, n)); lam = rng.standard_normal((cnt, n))
    colmajor = np.ascontiguousarray(np.swapaxes(V, 1, 2))
    din, dout, dl = Dev(colmajor), Dev(shape=(cnt, n, n)), Dev(lam)
    check(lib.cuadmm_op_mul_diag_batch(dout.ptr, din.ptr, dl.ptr, n, cnt, None))
    got = np.swapaxes(dout.get(), 1, 2)
    assert np.array_equal(got, V * lam[:, None, :])


@pytest.mark.parametrize("n,cnt", [(2, 3), (16, 5), (32, 40), (45, 3), (100, 2), (7, 11)])
def test_mul_trans_batch_mfma(n, cnt):
    """P = T * V^T on the fp64 matrix cores (cublasDgemmStridedBatched(N,T), cublas.h:18-35), asymmetric inputs."""
    rng = np.random.default_rng(n)
    T = rng.standard_normal((cnt, n, n)); V = rng.standard_normal((cnt, n, n))
    cm = lambda a: np.ascontiguousarray(np.swapaxes(a, 1, 2))
    dT, dV, dP = Dev(cm(T)), Dev(cm(V)), Dev(shape=(cnt, n, n))
    check(lib.cuadmm_op_mul_trans_batch(dP.ptr, dT.ptr, dV.ptr, n, cnt, None))
    got = np.swapaxes(dP.get(), 1, 2)
    ref = T @ np.swapaxes(V, 1, 2)
    assert np.max(np.abs(got - ref)) <= 1e-13 * n * max(1, np.abs(ref).max())
    if n == 2:                                                            # test/cublas_test.hpp:3-40 style: M*M^T
        M = np.array([[1.0, 2.0], [3.0, 4.0]])
        dM, dQ = Dev(cm(M[None])), Dev(shape=(1, 2, 2))
        check(lib.cuadmm_op_mul_trans_batch(dQ.ptr, dM.ptr, dM.ptr, 2, 1, None))
        assert np.array_equal(np.swapaxes(dQ.get(), 1, 2)[0], M @ M.T)


def test_spmv_csr_and_axpby_and_norm():
    rng = np.random.default_rng(3)
    A = sp.random(300, 500, density=0.03, random_state=4, format="csr")
    x, y = rng.standard_normal(500), rng.standard_normal(300)
    drp, dci, dv = Dev(A.indptr.astype(np.int32)), Dev(A.indices.astype(np.int32)), Dev(A.data)
    dx, dy = Dev(x), Dev(y)
    check(lib.cuadmm_op_spmv_csr(300, drp.ptr, dci.ptr, dv.ptr, dx.ptr, dy.ptr, -1.0, 0.0, None))   # solver.cu:478
    assert np.max(np.abs(dy.get() + A @ x)) <= 1e-13
    dy = Dev(y)
    check(lib.cuadmm_op_spmv_csr(300, drp.ptr, dci.ptr, dv.ptr, dx.ptr, dy.ptr, 2.0, 0.5, None))
    assert np.max(np.abs(dy.get() - (2 * (A @ x) + 0.5 * y))) <= 1e-13
    a, b = rng.standard_normal(10001), rng.standard_normal(10001)
    da, db = Dev(a), Dev(b)
    check(lib.cuadmm_op_axpby2(da.ptr, db.ptr, 1.0, 0.7, a.size, None))          # X = X + tau*sig*Rd, solver.cu:758
    assert np.max(np.abs(da.get() - (a + 0.7 * b))) <= 1e-15 * 4
    dc = Dev(shape=(a.size,))
    check(lib.cuadmm_op_axpby3(dc.ptr, da.ptr, db.ptr, 0.25, -1.0, a.size, None))  # S = Xdiff/sig - Rd1, solver.cu:656
    assert np.max(np.abs(dc.get() - (0.25 * da.get() - b))) <= 1e-15 * 4
    nrm = C.c_double()
    check(lib.cuadmm_op_norm2(db.ptr, b.size, C.byref(nrm), None))
    assert abs(nrm.value - np.linalg.norm(b)) <= 1e-13 * np.linalg.norm(b)


def test_cli_matches_engine_and_reference_format(problem_dirs, ref_logs):
    """cuadmm_exe <dir/> (src/main.cu:8-44): console census/table and X_opt.txt."""
    exe = os.path.join(os.path.dirname(cuadmm_amd.LIB_PATH), "cuadmm_exe")
    d = problem_dirs["hinf12"]
    r = subprocess.run([exe, d, "--max_iter=60", "--switch_admm=5000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert "Loaded problem from " + d in out and "Analysis of the blk vector:" in out
    assert "  it. | p infeas d infeas | primal obj.   dual obj. rel. gap |  time |   sigma | " in out
    assert "Solver ended: maximum iteration reached" in out and "time per iteration" in out
    rows = [ln for ln in out.splitlines() if ln.startswith("    0 |") or ln.startswith("   50 |") or ln.startswith("   60 |")]
    assert len(rows) == 3
    p = orc.load_problem_txt(d)
    o = orc.OracleSolver().init_problem(p)
    o.solve(60, 1e-3, 0, 50, 100, 5000, 1.05)
    want50 = orc.LOG_ROW_FMT % (50, o.info.errRp[49], o.info.errRd[49], o.info.pobj[49], o.info.dobj[49], o.info.relgap[49], 0.0, o.info.sig[49])
    got = [ln for ln in rows if ln.startswith("   50 |")][0]
    assert got.split("|")[1:3] == want50.split("|")[1:3] and got.split("|")[4] == want50.split("|")[4]
    X = np.array([float(x) for x in open(d + "X_opt.txt").read().split()])
    assert X.size == p.vec_len and np.max(np.abs(X - o.X)) <= 1e-8 * (1 + np.max(np.abs(o.X)))
    first = open(d + "X_opt.txt").readline().rstrip("\n")
    assert len(first.split(".")[1]) == 32                                           # "%.32f"
    # unreadable directory -> exit(1) like the reference (io.cu:30-33)
    r2 = subprocess.run([exe, "/nonexistent/"], capture_output=True, text=True)
    assert r2.returncode == 1
    # --json=<file> (not in the reference): the run's figures beside the console table, only when asked for
    import json, tempfile
    with tempfile.TemporaryDirectory() as tmp:
        side = os.path.join(tmp, "run.json")
        r3 = subprocess.run([exe, d, "--max_iter=60", "--switch_admm=5000", "--quiet", "--json=" + side], capture_output=True, text=True, timeout=300)
        assert r3.returncode == 0, r3.stderr
        j = json.load(open(side))
        assert j["iterations"] == 60 and j["vec_len"] == p.vec_len and j["con_num"] == p.con_num
        assert abs(j["pobj"] - o.info.pobj[59]) <= 1e-8 * (1 + abs(o.info.pobj[59]))
        assert j["iters_per_s"] > 0 and j["phases"]["psd_project"]["launches"] >= 60 and j["psd_project_nominal_tflops"] > 0


@pytest.mark.parametrize("n", [64, 192, 1024])
def test_gemm_sym_op_fp64_mfma(n):
    """The DGEMM behind the large-block projection (psd_large.hip) against numpy; A must be symmetric."""
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n)); A = (A + A.T) / 2
    B = rng.standard_normal((n, n))
    E = rng.standard_normal((n, n))
    dA, dB, dE, dC = Dev(A), Dev(B), Dev(E), Dev(shape=(n, n))
    check(lib.cuadmm_op_gemm_sym(n, dA.ptr, dB.ptr, 0.75, -1.25, dE.ptr, dC.ptr, None))
    check(lib.cuadmm_dev_sync())
    ref = 0.75 * A @ B - 1.25 * E
    assert np.max(np.abs(dC.get() - ref)) <= 1e-13 * n
    check(lib.cuadmm_op_gemm_sym(n, dA.ptr, dB.ptr, 1.0, 0.0, None, dC.ptr, None))
    check(lib.cuadmm_dev_sync())
    assert np.max(np.abs(dC.get() - A @ B)) <= 1e-13 * n
    with pytest.raises(cuadmm_amd.CuadmmError):
        check(lib.cuadmm_op_gemm_sym(100, dA.ptr, dB.ptr, 1.0, 0.0, None, dC.ptr, None))


@pytest.mark.parametrize("k", [1, 50, 64, 100, 192, 1000, 2500])
def test_tail_solve_op_vs_triangular_solves(k):
    """z <- L^-T D^-1 L^-1 z with inv(L) from recursive doubling on the matrix cores (tail_solve.hip), ragged sizes."""
    import scipy.linalg as sl
    rng = np.random.default_rng(k)
    L = np.tril(rng.standard_normal((k, k)) * (0.5 / np.sqrt(k)), -1) + np.eye(k)
    D = rng.uniform(0.1, 2.0, k) * rng.choice([1.0, 1.0, 1.0, -1.0], k)       # LDL^T pivots may be negative
    z = rng.standard_normal((3, k))
    ref = np.stack([sl.solve_triangular(L.T, sl.solve_triangular(L, zi, lower=True, unit_diagonal=True) / D,
                                        lower=False, unit_diagonal=True) for zi in z])
    got = z.copy()
    check(lib.cuadmm_op_tail_solve(L.ctypes.data_as(C.c_void_p), D.ctypes.data_as(C.c_void_p), k, got.ctypes.data_as(C.c_void_p), 3))
    assert np.linalg.norm(got - ref) <= 1e-13 * np.linalg.norm(ref)


@pytest.mark.parametrize("k", [18500, 24700, 33000])
def test_tail_solve_op_beyond_one_workgroups_reach(k):
    """K > 18 432: four workgroups share a row of inv(L) and exchange their parts of u = W z through sentinel slots
    (ts_onepass_group_kernel: 6 columns per thread up to 24 576, 8 beyond; EIGHT workgroups per row beyond 32 768 columns -- round 5,
    what option tail_max_k admits); three solves in a row, so the slots are reset and reused.
    Same tolerance as the small sizes; the second and third right-hand sides repeat the first, and must reproduce it bit for bit."""
    import scipy.linalg as sl
    rng = np.random.default_rng(k)
    L = rng.random((k, k), dtype=np.float32).astype(np.float64)
    L -= 0.5
    L *= 1.0 / np.sqrt(k)
    L = np.tril(L, -1)
    L[np.diag_indices(k)] = 1.0
    D = rng.uniform(0.1, 2.0, k) * rng.choice([1.0, 1.0, 1.0, -1.0], k)
    z0 = rng.standard_normal(k)
    ref = sl.solve_triangular(L.T, sl.solve_triangular(L, z0, lower=True, unit_diagonal=True) / D, lower=False, unit_diagonal=True)
    got = np.stack([z0, z0, z0]).copy()
    check(lib.cuadmm_op_tail_solve(L.ctypes.data_as(C.c_void_p), D.ctypes.data_as(C.c_void_p), k, got.ctypes.data_as(C.c_void_p), 3))
    assert np.linalg.norm(got[0] - ref) <= 1e-13 * np.linalg.norm(ref)
    assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2])


@pytest.mark.parametrize("k,world,one_pass", [(50, 8, 1), (50, 8, 0), (1000, 3, 1), (1000, 7, 0), (10240, 8, 1), (10240, 5, 0), (18500, 8, 1),
                                              (50, 8, 3), (50, 8, 2), (1000, 3, 3), (5000, 4, 2), (5000, 8, 3), (18500, 2, 3)])
def test_tail_solve_sharded_partials_sum_to_the_solve(k, world, one_pass):
    """The dense tail as the ranks of a sharded engine apply it (TailSolve::shard_*, cuadmm_tail_shard_bounds): rank p takes the rows
    of its share of the triangle and the partial results are summed by the all-reduce.  All `world` partials from one process: their
    sum is the solve (1e-13 against scipy), rows_out are the bounds' differences, and a rank with an EMPTY range (k = 50 on eight
    ranks: rows [8, 8)) contributes exact zeros -- with the one-pass kernels, the row-sharing kernel (k > 18 432) and the two
    triangular GEMVs alike.  one_pass + 2: every rank keeps ONLY its rows of inv(L22) (TailSolve::keep_shard: a compact matrix at the
    rows' own width, W^T gone -- the fallback's second pass accumulates by columns); same partials."""
    import scipy.linalg as sl
    rng = np.random.default_rng(k + world)
    L = rng.random((k, k), dtype=np.float32).astype(np.float64)
    L -= 0.5
    L *= 1.0 / np.sqrt(k)
    L = np.tril(L, -1)
    L[np.diag_indices(k)] = 1.0
    D = rng.uniform(0.1, 2.0, k) * rng.choice([1.0, 1.0, 1.0, -1.0], k)
    z = rng.standard_normal(k)
    ref = sl.solve_triangular(L.T, sl.solve_triangular(L, z, lower=True, unit_diagonal=True) / D, lower=False, unit_diagonal=True)
    out = np.full((world, k), np.nan)
    rows = np.zeros(world, np.int32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    check(lib.cuadmm_op_tail_solve_sharded(P(L), P(D), k, P(z), world, one_pass, P(out), P(rows)))
    b = np.zeros(world + 1, np.int32)
    check(lib.cuadmm_tail_shard_bounds(k, world, P(b)))
    assert np.array_equal(rows, np.diff(b))
    assert np.all(np.isfinite(out))
    for p in range(world):
        if rows[p] == 0:
            assert not out[p].any(), p                   # nothing to apply: zeros, not garbage
    if k == 50:
        assert np.count_nonzero(rows == 0) >= 1
    assert np.linalg.norm(out.sum(axis=0) - ref) <= 1e-13 * np.linalg.norm(ref)


def test_tail_solve_lost_exchange_protocol():
    """What happens when the row-sharing kernel cannot rely on co-residency (HIP does not promise it): a NaN with the sentinel's
    bits in the right-hand side does not stall the exchange (it is canonicalised before it is published); a raised failure
    counter makes the next launch return NaN at once instead of spending seconds per row round; take_failure() reports and clears
    it and the object continues on the two triangular GEMVs -- with the right answer."""
    import time
    import scipy.linalg as sl
    k = 18500
    rng = np.random.default_rng(5)
    L = rng.random((k, k), dtype=np.float32).astype(np.float64)
    L -= 0.5
    L *= 1.0 / np.sqrt(k)
    L = np.tril(L, -1)
    L[np.diag_indices(k)] = 1.0
    D = rng.uniform(0.1, 2.0, k)
    z = rng.standard_normal(k)
    ref = sl.solve_triangular(L.T, sl.solve_triangular(L, z, lower=True, unit_diagonal=True) / D, lower=False, unit_diagonal=True)
    out = np.zeros((4, k))
    counts = np.zeros(3, np.int32)
    t0 = time.time()
    check(lib.cuadmm_op_tail_solve_drill(L.ctypes.data_as(C.c_void_p), D.ctypes.data_as(C.c_void_p), k, z.ctypes.data_as(C.c_void_p),
                                         out.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p)))
    assert time.time() - t0 < 60.0                       # build + four solves; one lost round alone used to cost ~4 s, sixty per solve
    assert np.linalg.norm(out[0] - ref) <= 1e-13 * np.linalg.norm(ref)
    assert np.isnan(out[1]).any() and counts[0] == 0     # poisoned right-hand side: NaN out, no exchange lost
    assert np.isnan(out[2]).all() and counts[1] >= 1     # counter raised beforehand: every workgroup leaves NaN and returns
    assert counts[2] == 1
    assert np.linalg.norm(out[3] - ref) <= 1e-13 * np.linalg.norm(ref)


@pytest.mark.parametrize("k", [1, 50, 64, 130, 1000, 2000])
def test_tail_factor_solve_op_dense_ldlt_on_gpu(k):
    """Dense LDL^T (no pivoting) + inverse + two GEMVs on the GPU for a Schur complement given as sparse lower triangle
    (what cuadmm_aat_create_split hands over); indefinite-but-factorable matrices included (LDL^T pivots of both signs)."""
    rng = np.random.default_rng(k + 7)
    G = rng.standard_normal((k, k)) / np.sqrt(k)
    S = G @ G.T + 0.5 * np.eye(k)
    if k >= 50:                                  # make the trailing half negative definite: pivots change sign
        J = np.ones(k); J[k // 2:] = -1.0
        S = (S * J) * J[:, None] * 1.0
        S[k // 2:, k // 2:] *= -1.0
    Sl = sp.csr_matrix(np.tril(S))
    rp, ci, vv = Sl.indptr.astype(np.int64), Sl.indices.astype(np.int32), Sl.data.astype(np.float64)
    z = rng.standard_normal((2, k))
    ref = np.linalg.solve(S, z.T).T
    got = z.copy()
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    check(lib.cuadmm_op_tail_factor_solve(P(rp), P(ci), P(vv), k, P(got), 2))
    assert np.linalg.norm(got - ref) <= 1e-11 * np.linalg.norm(ref)
    # a zero pivot is reported like the host factor does ("Factorization fails!")
    Z = sp.csr_matrix(np.tril(np.zeros((3, 3)) + np.diag([1.0, 0.0, 1.0])))
    with pytest.raises(cuadmm_amd.CuadmmError) as e:
        zz = np.ones(3)
        check(lib.cuadmm_op_tail_factor_solve(P(Z.indptr.astype(np.int64)), P(Z.indices.astype(np.int32)), P(Z.data.astype(np.float64)), 3, P(zz), 1))
    assert "Factorization fails" in str(e.value)


@pytest.mark.parametrize("nbytes", [8, 4096, (16 << 20) - 8, 16 << 20, (16 << 20) + 8, 40 << 20])
def test_staged_copies_round_trip(nbytes):
    """cuadmm_memcpy_h2d / _d2h go through the library's page-locked staging halves (csrc/staging.hip: 2 x 16 MB) -- never
    hipMemcpy on caller memory: sizes around the chunk boundaries, freshly allocated and freed host arrays in between (the
    pattern that made the runtime's cached registrations go stale)."""
    rng = np.random.default_rng(nbytes % 1000)
    n = nbytes // 8
    for rep in range(3):
        x = rng.standard_normal(n)
        want = x.copy()
        d = Dev(x)
        del x                                   # the source array is gone before the next allocation / copy
        junk = np.full(n + 1024 * rep, float(rep))   # the freed range goes to another owner
        y = d.get()
        assert y.shape == (n,) and np.array_equal(y, want)
        del junk, d
