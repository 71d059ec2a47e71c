"""CPU tests of the C-ABI shared library: it loads, exports every symbol declared in
include/cuadmm_amd.h, and its host-side entry points (loader, block bookkeeping, AA^T factor)
agree with the oracle and with the reference's unit-test vectors.  No device compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

import cuadmm_amd
from cuadmm_amd._lib import PROTOTYPES, check
from oracle import cuadmm_oracle as orc
from tests.conftest import GOLDEN, ROOT

lib = cuadmm_amd.load()
P = lambda a: a.ctypes.data_as(C.c_void_p)


def test_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "cuadmm_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(cuadmm_[a-z0-9_A-Z]+)\s*\(", hdr)) - {"cuadmm_allreduce_fn"}
    assert len(names) > 50
    raw = C.CDLL(cuadmm_amd.LIB_PATH)
    for n in sorted(names):
        getattr(raw, n)
    assert names == set(PROTOTYPES)                                        # the Python binding covers the header


def test_no_device_is_a_loud_error():
    if lib.cuadmm_device_count() > 0:
        pytest.skip("GPU present")
    s = cuadmm_amd.SDPSolver(verbose=False)
    with pytest.raises(cuadmm_amd.CuadmmError) as e:
        s.init(15, 30, 1, 1, [0, 1], [0], [1.0], 1, [0], [1.0], 1, [0], [1.0], 1, [1], 1)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)
    p = C.c_void_p()
    assert lib.cuadmm_dev_malloc(C.byref(p), 64) == -2


def test_maps_match_reference_kats():
    blk = np.array([5, 4], np.int32)
    mB, m1, m2 = (np.zeros(25, np.int32) for _ in range(3))
    check(lib.cuadmm_get_maps_duo(P(blk), 2, 5, 4, 25, P(mB), P(m1), P(m2)))
    oB, o1, o2 = orc.get_maps_duo([5, 4], 5, 4)
    assert mB.tolist() == oB.tolist() and m1.tolist() == o1.tolist() and m2.tolist() == o2.tolist()
    assert m1.tolist()[:6] == [0, 5, 6, 10, 11, 12] and m2.tolist()[:6] == [0, 1, 6, 2, 7, 12]   # utils_test.hpp:36-61


@pytest.mark.parametrize("blk", [[2, 4], [6] * 40 + [4], [7] * 5 + [10] * 12 + [28] * 3 + [55] * 2 + [120], [3, 45, 3, 45, 10],
                                 [1] * 30 + [2] * 5 + [33]])
def test_get_maps_vs_oracle(blk):
    blk = np.array(blk, np.int32)
    L = int(orc.svec_block_offsets(blk)[-1])
    mB, m1, m2 = (np.zeros(L, np.int32) for _ in range(3))
    check(lib.cuadmm_get_maps(P(blk), blk.size, L, P(mB), P(m1), P(m2)))
    sizes, nums = orc.analyze_blk(blk)
    oB, o1, o2 = orc.get_maps(blk, orc.MatrixSizes(sizes, nums))
    assert np.array_equal(mB, oB) and np.array_equal(m1, o1) and np.array_equal(m2, o2)    # bit-exact indexing
    s_out, n_out = np.zeros(64, np.int32), np.zeros(64, np.int32)
    k = lib.cuadmm_analyze_blk(P(blk), blk.size, P(s_out), P(n_out), 64)
    assert s_out[:k].tolist() == sizes and n_out[:k].tolist() == nums
    for s_, n_ in zip(sizes, nums):
        assert bool(lib.cuadmm_is_large_mat(s_, n_)) == orc.is_large_mat(s_, n_)


def test_get_maps_kat_two_sizes():
    # test/kernels_test.hpp:339-341: blk = {2,4} -> 2 is "small", 4 is "small" too under the heuristic, so use
    # the duo variant that the KAT was generated with (LARGE=4, SMALL=2)
    blk = np.array([2, 4], np.int32)
    mB, m1, m2 = (np.zeros(13, np.int32) for _ in range(3))
    check(lib.cuadmm_get_maps_duo(P(blk), 2, 4, 2, 13, P(mB), P(m1), P(m2)))
    assert mB.tolist() == [1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]
    assert m1.tolist() == [0, 2, 3, 0, 4, 5, 8, 9, 10, 12, 13, 14, 15]
    assert m2.tolist() == [0, 1, 3, 0, 1, 5, 2, 6, 10, 3, 7, 11, 15]


def test_inverse_permutation_kat():
    perm = np.array([10, 6, 2, 4, 0, 8, 1, 3, 5, 7, 9], np.int32)              # test/utils_test.hpp:8-17
    inv = np.zeros(11, np.int32)
    check(lib.cuadmm_inverse_permutation(P(perm), 11, P(inv)))
    assert all(perm[inv[i]] == i for i in range(11))
    bad = np.array([0, 5], np.int32)
    assert lib.cuadmm_inverse_permutation(P(bad), 2, P(inv)) == -1


def test_coo_to_csc_and_blk_files():
    r, c, v = orc.read_coo(os.path.join(GOLDEN, "io", "sparse_matrix_coo.txt"))
    cp = np.zeros(5, np.int32)
    check(lib.cuadmm_coo_to_csc(P(cp), P(c), P(r), P(v), 6, 4))
    assert cp.tolist() == [0, 2, 4, 5, 6] and r.tolist() == [0, 2, 1, 3, 2, 2] and v.tolist() == [10, 30, 20, 60, 40, 50]
    # empty leading column (the case the reference conversion gets wrong)
    r2, c2, v2 = np.array([0, 1], np.int32), np.array([2, 1], np.int32), np.array([1.0, 2.0])
    cp2 = np.zeros(4, np.int32)
    check(lib.cuadmm_coo_to_csc(P(cp2), P(c2), P(r2), P(v2), 2, 3))
    assert cp2.tolist() == [0, 0, 1, 2] and r2.tolist() == [1, 0]
    types, sizes = (C.c_char * 8)(), np.zeros(8, np.int32)
    n = lib.cuadmm_read_blk(os.path.join(GOLDEN, "io", "blk_types.txt").encode(), types, P(sizes), 8)
    assert n == 3 and bytes(types[:3]) == b"abc" and sizes[:3].tolist() == [10, 20, 30]
    n = lib.cuadmm_read_blk(os.path.join(GOLDEN, "io", "blk_normal.txt").encode(), types, P(sizes), 8)
    assert n == 3 and bytes(types[:3]) == b"sss"
    assert lib.cuadmm_read_blk(b"/nonexistent/blk.txt", types, P(sizes), 8) == -3


@pytest.mark.parametrize("name", ["hinf12", "truss5", "rose13", "ros_2000", "cnhil10"])
def test_problem_from_txt_matches_oracle_loader(name, problem_dirs, capfd):
    p = cuadmm_amd.Problem.from_txt(problem_dirs[name])
    o = orc.load_problem_txt(problem_dirs[name])
    assert (p.vec_len, p.con_num, p.mat_num, p.At_nnz) == (o.vec_len, o.con_num, o.blk.size, o.At_nnz)
    assert np.array_equal(p.At_csc_col_ptrs, o.At_col_ptrs) and np.array_equal(p.At_csc_row_ids, o.At_row_ids)
    assert np.array_equal(p.At_csc_vals, o.At_vals)
    assert np.array_equal(p.b_indices, o.b_idx) and np.array_equal(p.b_vals, o.b_vals)
    assert np.array_equal(p.C_indices, o.C_idx) and np.array_equal(p.C_vals, o.C_vals)
    out = capfd.readouterr().out
    assert "Loaded problem from" in out and ("vector length: %d" % o.vec_len) in out     # problem.cu:74-80


def test_problem_from_txt_errors(tmp_path):
    with pytest.raises(cuadmm_amd.CuadmmError) as e:
        cuadmm_amd.Problem.from_txt(str(tmp_path) + "/")
    assert e.value.code == -3
    d = tmp_path / "p"; d.mkdir()
    (d / "blk.txt").write_text("q 3\n"); (d / "con_num.txt").write_text("1\n")      # 'u n' is accepted since round 2 (test_f4_free_and_rank.py)
    for f in ("At.txt", "b.txt", "C.txt"):
        (d / f).write_text("")
    with pytest.raises(cuadmm_amd.CuadmmError) as e:
        cuadmm_amd.Problem.from_txt(str(d) + "/")
    assert "unknown block type 'q'" in str(e.value)                               # problem.cu:33-35


def _aat(p):
    At = sp.csc_matrix((p.At_vals, p.At_row_ids, p.At_col_ptrs), shape=(p.vec_len, p.con_num))
    A = At.T.tocsc(); A.sort_indices()
    h = C.c_void_p()
    cp, ri, vx = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    check(lib.cuadmm_aat_create(p.con_num, p.vec_len, P(cp), P(ri), P(vx), 1e-15, C.byref(h)))
    return h, A


@pytest.mark.parametrize("name", ["hinf12", "truss5", "rose13", "ros_2000"])
def test_aat_factor_and_permuted_solve(name, problem_dirs):
    p = orc.load_problem_txt(problem_dirs[name])
    m = p.con_num
    h, A = _aat(p)
    perm = np.ctypeslib.as_array(lib.cuadmm_aat_perm(h), shape=(m,)).copy()
    assert sorted(perm.tolist()) == list(range(m))
    rng = np.random.default_rng(0)
    rhs = A @ rng.standard_normal(p.vec_len)                                       # consistent right-hand side
    # contract of cholmod_solve2(CHOLMOD_LDLt): rhs_perm[i] = rhs[perm[i]], y[perm[i]] = sol_perm[i]
    rp = np.ascontiguousarray(rhs[perm]); sol = np.empty(m)
    check(lib.cuadmm_aat_solve_permuted(h, P(rp), P(sol)))
    y = np.empty(m); y[perm] = sol
    B = (A @ A.T) + 1e-15 * sp.identity(m)
    assert np.linalg.norm(B @ y - rhs) <= 1e-10 * np.linalg.norm(rhs)
    # without the permutation the answer is wrong (documents the contract, cholesky_cpu_test.hpp:57-100)
    if name != "ros_2000" and not np.array_equal(perm, np.arange(m)):
        sol2 = np.empty(m)
        check(lib.cuadmm_aat_solve_permuted(h, P(np.ascontiguousarray(rhs)), P(sol2)))
        assert np.linalg.norm(B @ sol2 - rhs) > 1e-6 * np.linalg.norm(rhs)
    assert lib.cuadmm_aat_factor_nnz(h) >= 0
    lib.cuadmm_aat_free(h)


def test_aat_all_ones_solution():
    # test/cholesky_cpu_test.hpp:3-55 style: A = I (4x4) scaled, solution of (AA^T) x = AA^T 1 is all ones
    A = sp.random(30, 80, density=0.15, random_state=3, format="csc") + sp.eye(30, 80, format="csc")
    A = A.tocsc(); A.sort_indices()
    h = C.c_void_p()
    cp, ri, vx = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    check(lib.cuadmm_aat_create(30, 80, P(cp), P(ri), P(vx), 1e-16, C.byref(h)))
    perm = np.ctypeslib.as_array(lib.cuadmm_aat_perm(h), shape=(30,)).copy()
    rhs = (A @ A.T) @ np.ones(30)
    sol = np.empty(30)
    check(lib.cuadmm_aat_solve_permuted(h, P(np.ascontiguousarray(rhs[perm])), P(sol)))
    assert np.max(np.abs(sol - 1.0)) <= 1e-5                                       # the reference's tolerance
    lib.cuadmm_aat_free(h)


def test_partition_blocks_balanced_and_contiguous():
    rng = np.random.default_rng(0)
    blk = rng.choice([3, 6, 10, 15, 28, 45], size=1000).astype(np.int32)
    for world in (1, 2, 4, 8):
        first = np.zeros(world + 1, np.int32)
        check(lib.cuadmm_partition_blocks(P(blk), blk.size, world, P(first)))
        assert first[0] == 0 and first[-1] == blk.size and np.all(np.diff(first) >= 0)
        cost = np.array([np.sum(blk[first[r]:first[r + 1]].astype(float) ** 3) for r in range(world)])
        assert cost.max() <= cost.sum() / world + 45.0 ** 3 + 1
    first = np.zeros(9, np.int32)
    check(lib.cuadmm_partition_blocks(P(np.array([5, 5], np.int32)), 2, 8, P(first)))     # more ranks than blocks
    assert first[-1] == 2 and np.all(np.diff(first) >= 0)


def test_write_dense_txt_format(tmp_path):
    v = np.array([1.0, -0.5, 1e-3])
    fn = str(tmp_path / "X_opt.txt")
    check(lib.cuadmm_write_dense_txt(fn.encode(), P(v), 3))
    lines = open(fn).read().splitlines()
    assert lines == ["%.32f" % x for x in v]                                        # memory.h:278-294


@pytest.mark.parametrize("name,frac", [("rose13", 0.3), ("truss5", 0.5), ("hinf12", 1.0)])
def test_aat_dense_tail_split_matches_full_solve(name, frac, problem_dirs):
    """The split used by the engine's GPU tail (aat_ldlt.cpp / tail_solve.hip): leading sparse columns on the host,
    trailing k x k dense unit-lower triangle solved separately (here by scipy), equals the one-piece solve."""
    import scipy.linalg as sl
    p = orc.load_problem_txt(problem_dirs[name])
    m = p.con_num
    h, A = _aat(p)
    k = max(1, int(m * frac))
    rng = np.random.default_rng(1)
    rhs = rng.standard_normal(m)
    ref = np.empty(m)
    check(lib.cuadmm_aat_solve_permuted(h, P(rhs), P(ref)))
    L22 = np.full((k, k + 3), np.nan); D2 = np.empty(k)                          # leading dimension > k
    check(lib.cuadmm_aat_tail_dense(h, k, P(L22), k + 3, P(D2)))
    L22 = L22[:, :k]
    assert np.all(np.diag(L22) == 1.0) and np.all(np.triu(L22, 1) == 0.0)
    Lp = np.ctypeslib.as_array(lib.cuadmm_aat_factor_colptr(h), shape=(m + 1,))
    assert np.count_nonzero(np.tril(L22, -1)) <= Lp[m] - Lp[m - k]
    x = rhs.copy()
    check(lib.cuadmm_aat_solve_leading_forward(h, k, P(x)))
    z2 = sl.solve_triangular(L22, x[m - k:], lower=True, unit_diagonal=True)
    x[m - k:] = sl.solve_triangular(L22.T, z2 / D2, lower=False, unit_diagonal=True)
    check(lib.cuadmm_aat_solve_leading_backward(h, k, P(x)))
    assert np.linalg.norm(x - ref) <= 1e-12 * np.linalg.norm(ref)
    # the cost model keeps small factors on the host
    assert lib.cuadmm_aat_tail_plan(h, 32768) == 0
    with pytest.raises(cuadmm_amd.CuadmmError):
        check(lib.cuadmm_aat_tail_dense(h, m + 1, P(L22), m + 1, P(D2)))
    lib.cuadmm_aat_free(h)


def test_aat_tail_plan_picks_the_dense_triangle():
    """PlanarHand_N=1 (BASELINE config 1 data): L has 13.5 M nonzeros, > 90 % of them in the last ~12 000 columns."""
    from tests.conftest import load_npz_problem
    p = load_npz_problem("pendulum_N=80")
    At = sp.csc_matrix((p.At_vals, p.At_row_ids, p.At_col_ptrs), shape=(p.vec_len, p.con_num))
    A = At.T.tocsc(); A.sort_indices()
    h = C.c_void_p()
    cp, ri, vx = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    check(lib.cuadmm_aat_create(p.con_num, p.vec_len, P(cp), P(ri), P(vx), 1e-15, C.byref(h)))
    m = p.con_num
    k = lib.cuadmm_aat_tail_plan(h, 32768)
    Lp = np.ctypeslib.as_array(lib.cuadmm_aat_factor_colptr(h), shape=(m + 1,))
    assert 0 < k <= 32768 and k % 256 == 0
    assert Lp[m] - Lp[m - k] >= 0.4 * Lp[m]                                      # the tail holds a large share of nnz(L)
    assert lib.cuadmm_aat_tail_plan(h, 128) == 0                                 # cap below the smallest candidate
    lib.cuadmm_aat_free(h)


@pytest.mark.parametrize("name,whole", [("taha1a", True), ("swissroll", True), ("rose13", False), ("truss5", False)])
def test_aat_small_systems_take_the_whole_factor_as_the_dense_tail(name, whole, problem_dirs):
    """plan_tail for 512 <= m <= 4096 (csrc/aat_ldlt.cpp): a host solve is its nonzeros + ~35 us of PCIe hops and synchronisation,
    the whole factor as a dense tail on the device ~40 us of launches + one pass over 8 m^2 bytes.  taha1a (m = 3 002, 159 k nonzeros in
    L) and swissroll (3 380, 101 k) go to the device entirely; rose13's A A^T is DIAGONAL (2 379 nonzeros: the device-side forest solve
    serves it) and truss5 (m = 208) is below the range: both keep the one-piece factor."""
    from tests.conftest import load_npz_problem
    p = orc.load_problem_txt(problem_dirs[name]) if name in problem_dirs else load_npz_problem(name)
    At = sp.csc_matrix((p.At_vals, p.At_row_ids, p.At_col_ptrs), shape=(p.vec_len, p.con_num))
    A = At.T.tocsc(); A.sort_indices()
    cp, ri, vx = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    hs = C.c_void_p()
    check(lib.cuadmm_aat_create_split(p.con_num, p.vec_len, P(cp), P(ri), P(vx), 1e-15, 32768, C.byref(hs)))
    assert lib.cuadmm_aat_tail_k(hs) == (p.con_num if whole else 0)
    lib.cuadmm_aat_free(hs)


@pytest.mark.parametrize("name,frac", [("rose13", 0.3), ("truss5", 0.5), ("hinf12", 1.0), ("ros_2000", 0.1)])
def test_aat_split_factor_schur_complement(name, frac, problem_dirs):
    """cuadmm_aat_create_split leaves the last k columns unfactored and hands over the dense Schur complement:
    it must equal L22 D2 L22^T of the one-piece factor, and the leading columns must be identical."""
    p = orc.load_problem_txt(problem_dirs[name])
    m = p.con_num
    h, A = _aat(p)
    k = max(1, int(m * frac))
    cp, ri, vx = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    hs = C.c_void_p()
    check(lib.cuadmm_aat_create_split(m, p.vec_len, P(cp), P(ri), P(vx), 1e-15, -k, C.byref(hs)))
    assert lib.cuadmm_aat_tail_k(hs) == k and lib.cuadmm_aat_tail_k(h) == 0
    assert lib.cuadmm_aat_factor_nnz(hs) == lib.cuadmm_aat_factor_nnz(h)
    assert np.array_equal(np.ctypeslib.as_array(lib.cuadmm_aat_perm(hs), shape=(m,)), np.ctypeslib.as_array(lib.cuadmm_aat_perm(h), shape=(m,)))
    rp, ci, vv = C.POINTER(C.c_int64)(), C.POINTER(C.c_int)(), C.POINTER(C.c_double)()
    check(lib.cuadmm_aat_tail_schur(hs, C.byref(rp), C.byref(ci), C.byref(vv)))
    rp = np.ctypeslib.as_array(rp, shape=(k + 1,)).copy()
    S = sp.csr_matrix((np.ctypeslib.as_array(vv, shape=(rp[-1],)).copy(), np.ctypeslib.as_array(ci, shape=(rp[-1],)).copy(), rp),
                      shape=(k, k)).toarray()
    L22 = np.empty((k, k)); D2 = np.empty(k)
    check(lib.cuadmm_aat_tail_dense(h, k, P(L22), k, P(D2)))
    ref = np.tril((L22 * D2) @ L22.T)
    assert np.max(np.abs(np.tril(S) - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref)))
    assert np.all(np.triu(S, 1) == 0.0)
    # leading sweeps agree bit for bit with the one-piece factor's leading sweeps
    rhs = np.random.default_rng(2).standard_normal(m)
    a, b = rhs.copy(), rhs.copy()
    check(lib.cuadmm_aat_solve_leading_forward(h, k, P(a))); check(lib.cuadmm_aat_solve_leading_forward(hs, k, P(b)))
    assert np.array_equal(a, b)
    check(lib.cuadmm_aat_solve_leading_backward(h, k, P(a))); check(lib.cuadmm_aat_solve_leading_backward(hs, k, P(b)))
    assert np.array_equal(a, b)
    # a split factor cannot run the one-piece solve
    with pytest.raises(cuadmm_amd.CuadmmError):
        check(lib.cuadmm_aat_solve_permuted(hs, P(rhs), P(a)))
    with pytest.raises(cuadmm_amd.CuadmmError):
        check(lib.cuadmm_aat_tail_dense(hs, k, P(L22), k, P(D2)))
    lib.cuadmm_aat_tail_schur_release(hs)
    with pytest.raises(cuadmm_amd.CuadmmError):
        check(lib.cuadmm_aat_tail_schur(hs, C.byref(C.POINTER(C.c_int64)()), C.byref(C.POINTER(C.c_int)()), C.byref(C.POINTER(C.c_double)())))
    lib.cuadmm_aat_free(hs); lib.cuadmm_aat_free(h)


def test_aat_split_factor_tail_rows_on_the_host_pool():
    """From 512 tail rows on, the tail rows of the up-looking factorisation run on the host pool in two phases (their rows of
    L21 against the leading factor, then their Schur rows) with the serial loop's arithmetic entry by entry: the Schur complement
    equals L22 D2 L22^T of the one-piece factor and the leading columns -- L21 included -- are the one-piece factor's."""
    from tests.conftest import load_npz_problem
    p = load_npz_problem("PushBox_N=30_MOMENT")
    m = p.con_num
    h, A = _aat(p)
    k = 1536
    cp, ri, vx = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    hs = C.c_void_p()
    check(lib.cuadmm_aat_create_split(m, p.vec_len, P(cp), P(ri), P(vx), 1e-15, -k, C.byref(hs)))
    assert lib.cuadmm_aat_tail_k(hs) == k and lib.cuadmm_host_pool_threads() > 1
    rp, ci, vv = C.POINTER(C.c_int64)(), C.POINTER(C.c_int)(), C.POINTER(C.c_double)()
    check(lib.cuadmm_aat_tail_schur(hs, C.byref(rp), C.byref(ci), C.byref(vv)))
    rp = np.ctypeslib.as_array(rp, shape=(k + 1,)).copy()
    S = sp.csr_matrix((np.ctypeslib.as_array(vv, shape=(rp[-1],)).copy(), np.ctypeslib.as_array(ci, shape=(rp[-1],)).copy(), rp), shape=(k, k)).toarray()
    L22 = np.empty((k, k)); D2 = np.empty(k)
    check(lib.cuadmm_aat_tail_dense(h, k, P(L22), k, P(D2)))
    ref = np.tril((L22 * D2) @ L22.T)
    assert np.max(np.abs(np.tril(S) - ref)) <= 1e-12 * max(1.0, np.max(np.abs(ref)))
    rhs = np.random.default_rng(4).standard_normal(m)
    a, b = rhs.copy(), rhs.copy()
    check(lib.cuadmm_aat_solve_leading_forward(h, k, P(a))); check(lib.cuadmm_aat_solve_leading_forward(hs, k, P(b)))
    assert np.max(np.abs(a - b)) <= 1e-12 * max(1.0, np.max(np.abs(a)))       # the sweeps read L21: its entries are the one-piece factor's
    lib.cuadmm_aat_free(hs); lib.cuadmm_aat_free(h)


def test_aat_split_factor_threaded_leading_sweeps_on_a_large_forest():
    """PushBox_N=30 (examples/SPOT/data/MOSEK, m = 154 256) at the HOST optimum of the tail size (10 240 columns, forced: the planner
    itself now takes 18 432, see the next test): the leading columns form ~5 000 trees, the deepest > 1 000 levels (too deep for the
    device-side sweeps), 1 M nonzeros -- the host sweeps run per chunk of trees on the host pool,
    tail updates through per-chunk accumulators added in chunk order.  Same result as the serial sweeps of the one-piece factor to
    roundoff (the tail's sums are associated differently), and the same bits call after call."""
    from tests.conftest import load_npz_problem
    p = load_npz_problem("PushBox_N=30_MOMENT")
    m = p.con_num
    h, A = _aat(p)
    cp, ri, vx = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    hs = C.c_void_p()
    check(lib.cuadmm_aat_create_split(m, p.vec_len, P(cp), P(ri), P(vx), 1e-15, -10240, C.byref(hs)))   # the host optimum, forced
    k = lib.cuadmm_aat_tail_k(hs)
    assert k == 10240
    rhs = np.random.default_rng(3).standard_normal(m)
    a, b, b2 = rhs.copy(), rhs.copy(), rhs.copy()
    check(lib.cuadmm_aat_solve_leading_forward(h, k, P(a)))       # one-piece factor asked for this split: the serial walk
    check(lib.cuadmm_aat_solve_leading_forward(hs, k, P(b)))
    check(lib.cuadmm_aat_solve_leading_forward(hs, k, P(b2)))
    assert np.array_equal(b, b2)
    assert np.array_equal(a[:m - k], b[:m - k])                   # leading entries: the same arithmetic per entry
    assert np.max(np.abs(a[m - k:] - b[m - k:])) <= 1e-12 * max(1.0, np.max(np.abs(a[m - k:])))
    b[m - k:] = a[m - k:]                                         # the same tail for the backward sweeps
    check(lib.cuadmm_aat_solve_leading_backward(h, k, P(a))); check(lib.cuadmm_aat_solve_leading_backward(hs, k, P(b)))
    assert np.array_equal(a, b)
    lib.cuadmm_aat_free(hs); lib.cuadmm_aat_free(h)


def _forest_stats(Lp, Li, m, k):
    """height and largest tree of the elimination forest over the leading m - k columns (parent = first sub-diagonal row of a column)"""
    n1 = m - k
    parent = np.full(m, -1, np.int64)
    nz = np.diff(Lp) > 0
    parent[nz] = Li[Lp[:-1][nz]]
    h = np.ones(n1, np.int64); sz = np.ones(n1, np.int64)
    height = big = 0
    for j in range(n1):
        pj = parent[j]
        if 0 <= pj < n1:
            h[pj] = max(h[pj], h[j] + 1); sz[pj] += sz[j]
        else:
            height = max(height, h[j]); big = max(big, sz[j])
    return int(height), int(big)


def test_aat_tail_plan_takes_a_larger_tail_when_it_makes_a_deep_forest_shallow():
    """plan_tail's deep-forest branch (csrc/aat_ldlt.cpp): PushBox_N=30's host optimum (k = 10 240) leaves a leading forest 1 135 levels
    deep -- sweeps on the host, two PCIe hops per solve.  A tail of 18 432 columns swallows the long chains (height 120, largest tree
    ~1 100 nodes: inside lead_solve.hip's LDS budget of 6 144), so the whole y-solve runs on the device (measured 3.66 -> 1.53 ms per
    sGS iteration; 18 688 and 19 456 columns, 67 levels, measure the same).  The planner must find that neighbourhood."""
    from tests.conftest import load_npz_problem
    p = load_npz_problem("PushBox_N=30_MOMENT")
    m = p.con_num
    h, A = _aat(p)
    Lp = np.ctypeslib.as_array(lib.cuadmm_aat_factor_colptr(h), shape=(m + 1,)).copy()
    Lpp = C.c_void_p(); Lip = C.c_void_p(); Lxp = C.c_void_p(); Dp = C.c_void_p()
    check(lib.cuadmm_aat_factor_arrays(h, C.byref(Lpp), C.byref(Lip), C.byref(Lxp), C.byref(Dp)))
    Li = np.ctypeslib.as_array(C.cast(Lip, C.POINTER(C.c_int)), shape=(int(Lp[-1]),)).copy()
    cp, ri, vx = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    hs = C.c_void_p()
    lib.cuadmm_aat_plan_allow_tops(0)                                # the planner of round 4 (engine: option lead_tops = 0)
    try:
        check(lib.cuadmm_aat_create_split(m, p.vec_len, P(cp), P(ri), P(vx), 1e-15, 32768, C.byref(hs)))
        k = lib.cuadmm_aat_tail_k(hs)
        assert k % 256 == 0 and 18432 <= k <= 20480 and lib.cuadmm_aat_tail_tops(hs) == 0
        height, big = _forest_stats(Lp, Li, m, k)
        assert height <= 256 and big <= 6144
        assert _forest_stats(Lp, Li, m, 10240)[0] > 1000                 # what the host optimum would have left
        # a cap below the shallow region keeps the host optimum
        hc = C.c_void_p()
        check(lib.cuadmm_aat_create_split(m, p.vec_len, P(cp), P(ri), P(vx), 1e-15, 16384, C.byref(hc)))
        assert lib.cuadmm_aat_tail_k(hc) == 10240
        lib.cuadmm_aat_free(hc); lib.cuadmm_aat_free(hs)
    finally:
        lib.cuadmm_aat_plan_allow_tops(1)
    # Round 5, dense tree tops (csrc/lead_solve.h): with the nodes of height >= 32 solved through explicit inverses the depth of the forest
    # stops deciding -- a tail BELOW the host optimum, the forest under it as deep as ever (measured 1.81 -> 0.99 ms per sGS iteration)
    ht = C.c_void_p()
    check(lib.cuadmm_aat_create_split(m, p.vec_len, P(cp), P(ri), P(vx), 1e-15, 32768, C.byref(ht)))
    kt = lib.cuadmm_aat_tail_k(ht)
    assert kt % 256 == 0 and 4096 <= kt <= 10240 and lib.cuadmm_aat_tail_tops(ht) == 32
    assert _forest_stats(Lp, Li, m, kt)[0] > 1000
    lib.cuadmm_aat_free(ht); lib.cuadmm_aat_free(h)


@pytest.mark.parametrize("name,k", [("PushBox_N=30_MOMENT", 10240), ("pendulum_N=80", 0)])
def test_aat_leading_sweeps_restricted_to_l11(name, k):
    """Hybrid y-solve (engine: lead_solve.h): the host sweeps L11 only, L21 products happen elsewhere (the GPU; numpy here).
    forward11 + (z2 = rhs2 - L21 z1) must reproduce the full forward sweep, backward11 with w = L21^T x2 the full backward sweep --
    threaded chunks (PushBox_N=30: 16 chunks of trees) and the serial walk (pendulum: below the chunking threshold) alike."""
    from tests.conftest import load_npz_problem
    p = load_npz_problem(name)
    m = p.con_num
    At = sp.csc_matrix((p.At_vals, p.At_row_ids, p.At_col_ptrs), shape=(p.vec_len, p.con_num))
    A = At.T.tocsc(); A.sort_indices()
    cp, ri, vx = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
    hs = C.c_void_p()
    check(lib.cuadmm_aat_create_split(m, p.vec_len, P(cp), P(ri), P(vx), 1e-15, -k if k else 32768, C.byref(hs)))
    k = lib.cuadmm_aat_tail_k(hs)
    assert k > 0
    n1 = m - k
    Lpp = C.c_void_p(); Lip = C.c_void_p(); Lxp = C.c_void_p(); Dp = C.c_void_p()
    check(lib.cuadmm_aat_factor_arrays(hs, C.byref(Lpp), C.byref(Lip), C.byref(Lxp), C.byref(Dp)))
    Lp = np.ctypeslib.as_array(C.cast(Lpp, C.POINTER(C.c_int64)), shape=(m + 1,))
    nnz = int(Lp[n1])
    Li = np.ctypeslib.as_array(C.cast(Lip, C.POINTER(C.c_int)), shape=(nnz,))
    Lx = np.ctypeslib.as_array(C.cast(Lxp, C.POINTER(C.c_double)), shape=(nnz,))
    Lfull = sp.csc_matrix((Lx, Li, np.concatenate([Lp[:n1 + 1], np.full(m - n1, Lp[n1])])), shape=(m, m))
    L21 = Lfull[n1:, :n1].tocsr()
    rhs = np.random.default_rng(5).standard_normal(m)
    full, part = rhs.copy(), rhs.copy()
    check(lib.cuadmm_aat_solve_leading_forward(hs, k, P(full)))
    check(lib.cuadmm_aat_solve_leading_forward11(hs, k, P(part)))
    assert np.array_equal(part[n1:], rhs[n1:])                                  # the tail part is not touched
    assert np.array_equal(part[:n1], full[:n1])                                 # the same arithmetic per leading entry
    z2 = rhs[n1:] - L21 @ part[:n1]
    assert np.max(np.abs(z2 - full[n1:])) <= 1e-12 * max(1.0, np.max(np.abs(full[n1:])))
    x2 = np.random.default_rng(6).standard_normal(k)
    full[n1:] = x2; part[n1:] = np.nan                                           # backward11 must not read the tail part
    w = np.ascontiguousarray(L21.T @ x2)
    check(lib.cuadmm_aat_solve_leading_backward(hs, k, P(full)))
    check(lib.cuadmm_aat_solve_leading_backward11(hs, k, P(part), P(w)))
    assert np.max(np.abs(part[:n1] - full[:n1])) <= 1e-11 * max(1.0, np.max(np.abs(full[:n1])))
    # a one-piece factor has no L21 to leave out
    h, _ = _aat(p)
    with pytest.raises(cuadmm_amd.CuadmmError):
        check(lib.cuadmm_aat_solve_leading_forward11(h, k, P(part)))
    lib.cuadmm_aat_free(h); lib.cuadmm_aat_free(hs)


def test_aat_threaded_solve_is_bitwise_identical_to_serial(tmp_path):
    """Large block-diagonal system (weak-scaled C2 structure, m = 200 000): the solve runs independent etree subtrees
    on the host pool; the result must not depend on the number of threads (bit for bit)."""
    import subprocess, sys, textwrap
    script = tmp_path / "solve.py"
    script.write_text(textwrap.dedent('''
        import sys, ctypes as C, numpy as np, scipy.sparse as sp
        sys.path.insert(0, %r)
        import cuadmm_amd
        from cuadmm_amd._lib import check
        lib = cuadmm_amd.load()
        P = lambda a: a.ctypes.data_as(C.c_void_p)
        rng = np.random.default_rng(5)
        nb, per, L = 40000, 5, 40000 * 30                      # 40 000 independent 5-constraint groups
        rows = np.repeat(np.arange(nb * per), 6)
        cols = (np.repeat(np.arange(nb), per * 6) * 30 + rng.integers(0, 30, nb * per * 6))
        A = sp.csc_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(nb * per, L)); A.sum_duplicates(); A.sort_indices()
        h = C.c_void_p()
        check(lib.cuadmm_aat_create(nb * per, L, P(A.indptr.astype(np.int32)), P(A.indices.astype(np.int32)), P(A.data), 1e-15, C.byref(h)))
        rhs = rng.standard_normal(nb * per); out = np.empty_like(rhs)
        check(lib.cuadmm_aat_solve_permuted(h, P(rhs), P(out)))
        perm = np.ctypeslib.as_array(lib.cuadmm_aat_perm(h), shape=(nb * per,))
        y = np.empty_like(out); y[perm] = out
        B = (A @ A.T).tocsr()
        r = np.empty_like(rhs); r[perm] = rhs
        print("RES", np.linalg.norm(B @ y - r) / np.linalg.norm(r))
        np.save(sys.argv[1], out)
    ''' % ROOT))
    outs = []
    for t in ("1", "6"):
        f = tmp_path / ("x%s.npy" % t)
        env = dict(os.environ, CUADMM_HOST_THREADS=t)
        r = subprocess.run([sys.executable, str(script), str(f)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert float(r.stdout.split("RES")[1]) < 1e-8
        outs.append(np.load(f))
    assert np.array_equal(outs[0], outs[1])


def test_host_pool_survives_two_concurrent_solvers():
    """Two factors with m >= 200 000 solved from two host threads at once (two engines in one process): the chunked solve
    of both reaches the process-wide host thread pool concurrently.  Round 1 deadlocked here (single-occupancy fork-join
    without an outer lock); run in a child process so that a regression is a timeout, not a hung test session."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import ctypes as C, threading, numpy as np, scipy.sparse as sp
        import cuadmm_amd
        lib = cuadmm_amd.load()
        P = lambda a: a.ctypes.data_as(C.c_void_p)
        m, L = 200000, 400000
        rng = np.random.default_rng(0)
        rows = np.repeat(np.arange(m), 2); cols = rng.integers(0, L, 2 * m)
        A = sp.csc_matrix((rng.standard_normal(2 * m), (rows, cols)), shape=(m, L)); A.sum_duplicates(); A.sort_indices()
        cp, ri, vx = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
        hs = []
        for _ in range(2):
            h = C.c_void_p()
            assert lib.cuadmm_aat_create(m, L, P(cp), P(ri), P(vx), 1e-15, C.byref(h)) == 0
            hs.append(h)
        rhs = rng.standard_normal(m); ref = np.empty(m)
        assert lib.cuadmm_aat_solve_permuted(hs[0], P(rhs), P(ref)) == 0
        outs = [np.empty(m), np.empty(m)]
        def work(i):
            for _ in range(30):
                assert lib.cuadmm_aat_solve_permuted(hs[i], P(rhs), P(outs[i])) == 0
        ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        [t.start() for t in ts]; [t.join() for t in ts]
        assert np.array_equal(outs[0], ref) and np.array_equal(outs[1], ref)
        print("OK")
    """)
    env = dict(os.environ, CUADMM_HOST_THREADS="6", PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


PICK_C1 = (9216, 16)


@pytest.mark.parametrize("name,m,nnzL,tail_k,perm_sha,tops_k,tops_cut", [
    ("PlanarHand_N=1_MOMENT", 66008, 13533205, 17152, "52f718bcc8f3773c", PICK_C1[0], PICK_C1[1]),
    ("pendulum_N=80", 112028, 929430, 10496, "53e3efdb51729949", 7168, 16),      # round 6: tree tops for pendulum too (0.647 -> 0.557 ms per sGS iteration)
    ("taha1a", 3002, 162451, 3002, "a76f23e6d0979eb3", 0, 0),
    ("PushBox_N=30_MOMENT", 154256, 2742805, 18688, "54923cf4d0953c4a", 7168, 32),
    ("PushT_N=30_MOMENT", 53290, 58473104, 27136, "51e27341946f6176", 15360, 16),
    ("bqp-r1-40-1", 269001, 816406, 0, "c02eb270d6b7a503", 1024, 32),      # round 5: the wide-forest plan (no dense tail pays; a small one in front of the device-side sweeps)
])
def test_ordering_and_tail_plan_of_the_fixtures_are_pinned(name, m, nnzL, tail_k, perm_sha, tops_k, tops_cut):
    """The fill-reducing ordering (own minimum degree with the near-clique exit -- confirmed by an EXACT degree count since round 5 --
    and the dense-row rule) and the tail planner decide nnz(L), the size of the GPU tail and with them every timing and tolerance
    stated for these inputs: a change to either must show up here first (the job of CHOLMOD's analyze in the reference,
    include/cuadmm/cholesky_cpu.h:62-140)."""
    import hashlib
    import scipy.sparse.linalg as spla
    from oracle import cuadmm_oracle as orc
    from tests.conftest import load_npz_problem
    p = load_npz_problem(name)
    real = orc.spla.factorized
    orc.spla.factorized = lambda M: None
    try:
        s = orc.OracleSolver().init_problem(p)
    finally:
        orc.spla.factorized = real
    At = s.At_csr
    L, mm = At.shape
    assert mm == m
    rp, ci, v = (np.ascontiguousarray(At.indptr, np.int32), np.ascontiguousarray(At.indices, np.int32), np.ascontiguousarray(At.data))
    h = C.c_void_p()
    lib.cuadmm_aat_factor_nnz.restype = C.c_int64
    # tail_k: the planner of round 4 (option lead_tops = 0); tops_k / tops_cut: the default since round 5 where the solve with dense tree tops
    # wins by the model (round 6: recalibrated on measured kernel times, the cut chosen from {16, 32}: profiles/r06_plan_picks.log), 0 where
    # the plan is unchanged
    lib.cuadmm_aat_plan_allow_tops(0)
    try:
        check(lib.cuadmm_aat_create_split(int(m), int(L), rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p),
                                          1e-15, 32768, C.byref(h)))
    finally:
        lib.cuadmm_aat_plan_allow_tops(1)
    try:
        perm = np.ctypeslib.as_array(lib.cuadmm_aat_perm(h), shape=(m,)).copy()
        assert sorted(perm.tolist()) == list(range(m))
        assert int(lib.cuadmm_aat_factor_nnz(h)) == nnzL
        assert int(lib.cuadmm_aat_tail_k(h)) == tail_k and lib.cuadmm_aat_tail_tops(h) == 0
        assert hashlib.sha256(perm.tobytes()).hexdigest()[:16] == perm_sha
    finally:
        lib.cuadmm_aat_free(h)
    h = C.c_void_p()
    check(lib.cuadmm_aat_create_split(int(m), int(L), rp.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p),
                                      1e-15, 32768, C.byref(h)))
    try:
        assert int(lib.cuadmm_aat_tail_k(h)) == (tops_k if tops_k else tail_k)
        assert lib.cuadmm_aat_tail_tops(h) == tops_cut
        assert np.array_equal(np.ctypeslib.as_array(lib.cuadmm_aat_perm(h), shape=(m,)), perm)      # the ordering does not depend on the plan
    finally:
        lib.cuadmm_aat_free(h)


def test_engine_options_are_validated_without_a_device():
    """cuadmm_set_option on a fresh handle touches no device: known keys are accepted, the one-pass kernel's ring depth is range-checked
    (csrc/engine.hip: tail_depth 0 ... 3), an unknown key is an error with a message."""
    import ctypes as C
    lib = cuadmm_amd.load()
    h = C.c_void_p()
    assert lib.cuadmm_create(C.byref(h)) == 0
    try:
        for key, val in (("tail_order", 0), ("tail_zreg", 0), ("tail_rb", 2), ("tail_depth", 3), ("psd_lg_clean", 1), ("tail_pivot", 1)):
            assert lib.cuadmm_set_option(h, key.encode(), C.c_double(val)) == 0, key
        assert lib.cuadmm_set_option(h, b"tail_depth", C.c_double(4)) != 0
        assert b"tail_depth" in lib.cuadmm_last_error()
        assert lib.cuadmm_set_option(h, b"no_such_option", C.c_double(1)) != 0
    finally:
        lib.cuadmm_destroy(h)
