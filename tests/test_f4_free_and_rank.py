"""unconstrained ('u') blocks and the rank-limited projection.

Reference status: `u n` is documented as work in progress (README.md:55-64) and its loader still rejects it
(src/problem.cu:28-36); the rank mask is prepared but switched off (src/duo_solver.cu:428-438, :843-850;
src/kernels/dense_scalar.cu:51-57; src/utils/get_eig_rank_mask.cu:13-37).  Here both are live: a 'u n' block is carried as
the negative size -n (n svec slots, identity "projection"), and eig_rank keeps the r largest eigenvalues of every PSD block.
The oracle restates both (oracle/cuadmm_oracle.py: blk_svec_len, psd_project_svec(eig_rank=...)); the GPU tests compare the
kernels and whole trajectories with it."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

import cuadmm_amd
from cuadmm_amd.synthetic import make_synthetic
from oracle import cuadmm_oracle as orc


def _write(dirname, name, text):
    with open(os.path.join(dirname, name), "w") as f:
        f.write(text)


def test_loader_maps_u_blocks_to_negative_sizes(tmp_path):
    d = str(tmp_path) + "/"
    _write(d, "blk.txt", "s 2\nu 3\n2\n")                    # svec: 3 + 3 + 3 = 9 slots
    _write(d, "con_num.txt", "2\n")
    _write(d, "At.txt", "0 0 1.0\n3 0 2.0\n8 1 -1.0\n")      # 0-based (row = svec slot, col = constraint)
    _write(d, "b.txt", "0 0 1.0\n")
    _write(d, "C.txt", "4 0 0.5\n")
    p = cuadmm_amd.Problem.from_txt(d)
    assert list(p.blk_vals) == [2, -3, 2] and p.vec_len == 9 and p.con_num == 2
    _write(d, "blk.txt", "s 2\nq 3\n")                       # any other letter stays an error (problem.cu:33-35)
    with pytest.raises(Exception):
        cuadmm_amd.Problem.from_txt(d)


def test_oracle_projection_with_free_blocks_and_rank_mask():
    rng = np.random.default_rng(0)
    blk = np.array([4, -3, 6, -1, 4])
    L = int(orc.blk_svec_len(blk).sum())
    assert L == 10 + 3 + 21 + 1 + 10
    x = rng.standard_normal(L)
    bidx = orc.BlockIndex(blk)
    out = orc.psd_project_svec(bidx, x)
    off = orc.svec_block_offsets(blk)
    assert np.array_equal(out[off[1]:off[2]], x[off[1]:off[2]]) and out[off[3]] == x[off[3]]      # free ranges untouched
    # rank mask = get_eig_rank_mask: last r of the ascending spectrum
    mask = np.zeros(2 * 6, int)
    for i in range(2):
        for j in range(2):
            mask[i * 6 + (6 - 1 - j)] = 1                                                         # get_eig_rank_mask.cu:30-35
    assert list(mask) == [0, 0, 0, 0, 1, 1] * 2
    r2 = orc.psd_project_svec(bidx, x, eig_rank=2)
    M = bidx.unpack(x)
    for (n, ids, ii, jj, gather), Mg in zip(bidx.groups, M):
        w, V = np.linalg.eigh(Mg)
        wp = np.maximum(w, 0) * np.r_[np.zeros(n - 2), np.ones(2)][None, :]
        P = (V * wp[:, None, :]) @ np.swapaxes(V, 1, 2)
        got = bidx.unpack(r2)[[g[0] for g in bidx.groups].index(n)]
        assert np.max(np.abs(got - P)) <= 1e-12
        assert np.all(np.linalg.matrix_rank(got, tol=1e-9) <= 2)


def _problem_with_free_block(seed=5, nfree=7):
    """strictly feasible primal / dual pair: PSD blocks from make_synthetic plus one free block that enters the constraints"""
    base = make_synthetic([6, 20, 33, 9, 70], cons_per_block=4, seed=seed)
    rng = np.random.default_rng(seed)
    L0, m = base.vec_len, base.con_num
    At0 = sp.csc_matrix((base.At_vals, base.At_row_ids, base.At_col_ptrs), shape=(L0, m))
    Au = sp.random(nfree, m, density=0.5, random_state=seed, format="csc", data_rvs=rng.standard_normal)
    At = sp.vstack([At0[:21], Au, At0[21:]]).tocsc(); At.sort_indices()     # the free block follows block 0 (6 x 6: 21 slots)
    blk = np.array([6, -nfree, 20, 33, 9, 70], np.int32)
    b = np.zeros(m); b[base.b_idx] = base.b_vals
    xu = rng.standard_normal(nfree)
    b = b + Au.T @ xu                                          # A [X0; xu]
    C0 = np.zeros(L0); C0[base.C_idx] = base.C_vals
    Cu = np.zeros(nfree)                                       # cost of the free variables
    C = np.concatenate([C0[:21], Cu, C0[21:]])
    bi, ci = np.nonzero(b)[0], np.nonzero(C)[0]
    return cuadmm_amd.Problem(L0 + nfree, m, blk, At.indptr, At.indices, At.data, bi, b[bi], ci, C[ci])


def _to_orc(p):
    return orc.Problem(p.vec_len, p.con_num, p.blk_vals, p.At_csc_col_ptrs, p.At_csc_row_ids, p.At_csc_vals,
                       p.b_indices, p.b_vals, p.C_indices, p.C_vals)


def test_oracle_admm_with_a_free_block_keeps_its_dual_slack_zero():
    p = _problem_with_free_block()
    o = orc.OracleSolver().init_problem(_to_orc(p))
    info = o.solve(60, 0.0, 0, 50, 100, 30, 1.05)
    off = orc.svec_block_offsets(p.blk_vals)
    assert np.max(np.abs(o.S[off[1]:off[2]])) <= 1e-12 * (1 + np.max(np.abs(o.S)))   # identity projection => S_u = 0
    assert info.errRp[-1] < info.errRp[0]


@pytest.mark.gpu
@pytest.mark.parametrize("eig_rank", [0, 1, 3])
def test_gpu_projection_free_blocks_and_rank_mask(eig_rank):
    from tests.helpers import psd_project_gpu
    rng = np.random.default_rng(3)
    blk = np.array([5, -4, 20, 32, -1, 50, 64, 3, 100, 16, 8, -9, 130, 150, 260], np.int32)   # 150, 260: the whole-chip eigensolver (eig_large.hip)
    L = int(orc.blk_svec_len(blk).sum())
    x = rng.standard_normal(L)
    ref = orc.psd_project_svec(orc.BlockIndex(blk), x, eig_rank=eig_rank)
    got = psd_project_gpu(x, blk, eig_rank=eig_rank)
    assert np.max(np.abs(got - ref)) <= 1e-11 * np.max(np.abs(x)) * 260
    off = orc.svec_block_offsets(blk)
    for k in np.nonzero(blk < 0)[0]:
        assert np.array_equal(got[off[k]:off[k + 1]], x[off[k]:off[k + 1]])


@pytest.mark.gpu
def test_gpu_rank_limited_projection_of_a_block_beyond_the_old_fence():
    """n = 1500 with a rank mask: refused in round 2 (one workgroup would have needed ~30 s); now one eigendecomposition on the
    whole chip per projection (csrc/eig_large.hip: eig_large_project)."""
    from tests.helpers import psd_project_gpu
    rng = np.random.default_rng(11)
    blk = np.array([1500, 7], np.int32)
    x = rng.standard_normal(int(orc.blk_svec_len(blk).sum()))
    ref = orc.psd_project_svec(orc.BlockIndex(blk), x, eig_rank=4)
    got = psd_project_gpu(x, blk, eig_rank=4)
    assert np.max(np.abs(got - ref)) <= 1e-11 * np.max(np.abs(x)) * 1500


@pytest.mark.gpu
def test_gpu_one_workgroup_kernels_small_plan_then_large_plan():
    """The cap on a kernel's dynamic LDS is per kernel and process-wide (hipFuncAttributeMaxDynamicSharedMemorySize): it is lifted
    once to everything the kernel's static LDS leaves (device_util.h: allow_max_dynamic_lds), never set to "what this plan needs" --
    a plan of small blocks followed (or accompanied) by one of large blocks on the same kernel must both launch."""
    from tests.helpers import psd_project_gpu
    rng = np.random.default_rng(12)
    for n in (70, 128, 66, 120):                                         # each call builds and drops a plan of its own
        blk = np.array([n], np.int32)
        x = rng.standard_normal(int(orc.blk_svec_len(blk).sum()))
        ref = orc.psd_project_svec(orc.BlockIndex(blk), x, eig_rank=5)
        got = psd_project_gpu(x, blk, eig_rank=5)
        assert np.max(np.abs(got - ref)) <= 1e-11 * np.max(np.abs(x)) * n


@pytest.mark.gpu
def test_gpu_solver_with_a_free_block_matches_the_oracle():
    p = _problem_with_free_block()
    s = cuadmm_amd.SDPSolver(verbose=False)
    s.init_problem(p)
    s.solve(40, 0.0, 0, 50, 100, 20, 1.05)
    o = orc.OracleSolver().init_problem(_to_orc(p))
    info = o.solve(40, 0.0, 0, 50, 100, 20, 1.05)
    for name, ref in (("pobj", info.pobj), ("dobj", info.dobj), ("errRp", info.errRp), ("errRd", info.errRd)):
        assert np.max(np.abs(s.info_arr(name) - np.asarray(ref)) / (1e-9 + np.abs(np.asarray(ref)))) <= 1e-7, name
    assert np.max(np.abs(s.X - o.X)) <= 1e-8 * (1 + np.max(np.abs(o.X)))
    off = orc.svec_block_offsets(p.blk_vals)
    assert np.max(np.abs(s.S[off[1]:off[2]])) <= 1e-12 * (1 + np.max(np.abs(s.S)))


@pytest.mark.gpu
def test_gpu_solver_rank_limited_projection_matches_the_oracle():
    base = make_synthetic([12, 30, 40, 7, 66], cons_per_block=4, seed=9)
    p = cuadmm_amd.Problem(base.vec_len, base.con_num, base.blk, base.At_col_ptrs, base.At_row_ids, base.At_vals,
                           base.b_idx, base.b_vals, base.C_idx, base.C_vals)
    s = cuadmm_amd.SDPSolver(verbose=False, eig_rank=3, eig_rank_begin_iter=6)      # full projection for 5 iterations, then rank 3
    s.init_problem(p)
    s.solve(25, 0.0, 0, 50, 100, 12, 1.05)
    o = orc.OracleSolver(eig_rank=3, eig_rank_begin_iter=6).init_problem(_to_orc(p))
    info = o.solve(25, 0.0, 0, 50, 100, 12, 1.05)
    for name, ref in (("pobj", info.pobj), ("dobj", info.dobj), ("errRp", info.errRp), ("errRd", info.errRd)):
        assert np.max(np.abs(s.info_arr(name) - np.asarray(ref)) / (1e-9 + np.abs(np.asarray(ref)))) <= 1e-7, name
    assert np.max(np.abs(s.X - o.X)) <= 1e-8 * (1 + np.max(np.abs(o.X)))
