"""GPU parity of the FUSED iteration (psd_fuse.h: the projection kernels of the 17 <= n <= 64 blocks form Xb themselves and
apply the S / X updates to their svec ranges) against the stand-alone kernels (CUADMM_FUSE=0) and the oracle.

Element by element the fused path evaluates the same expressions as aty_xb_kernel / post_kernel; only the order of the two
sums of an iteration differs.  Tolerance: per-iteration scalars <= 1e-9 relative over the run, sigma exact, X / y / S <= 1e-9.
"""
import numpy as np
import pytest

import cuadmm_amd
from cuadmm_amd import synthetic
from oracle import cuadmm_oracle as orc

pytestmark = pytest.mark.gpu


def _problem(blk, cons_per_block=3, seed=7):
    p = synthetic.make_synthetic(blk, cons_per_block=cons_per_block, seed=seed)
    return p, cuadmm_amd.Problem(p.vec_len, p.con_num, p.blk, p.At_col_ptrs, p.At_row_ids, p.At_vals, p.b_idx, p.b_vals, p.C_idx, p.C_vals)


def _coupled_problem(blk, n_couple, seed=5):
    """make_synthetic plus n_couple constraints whose nonzeros span TWO blocks: those rows are not local to a fused block
    and stay with the stand-alone SpMV (compact row list), next to the local ones evaluated inside the projection kernels."""
    p = synthetic.make_synthetic(blk, cons_per_block=2, seed=seed)
    rng = np.random.default_rng(seed)
    off = np.concatenate([[0], np.cumsum(np.asarray(p.blk, np.int64) * (np.asarray(p.blk, np.int64) + 1) // 2)])
    cp, rows, vals = list(p.At_col_ptrs), list(p.At_row_ids), list(p.At_vals)
    b_idx, b_vals = list(p.b_idx), list(p.b_vals)
    m = p.con_num
    for _ in range(n_couple):
        k1, k2 = rng.choice(len(p.blk), 2, replace=False)
        sl = sorted(set(int(off[k1] + rng.integers(0, off[k1 + 1] - off[k1])) for _ in range(3)) |
                    set(int(off[k2] + rng.integers(0, off[k2 + 1] - off[k2])) for _ in range(3)))
        rows += sl
        vals += list(rng.standard_normal(len(sl)))
        cp.append(len(rows))
        b_idx.append(m); b_vals.append(float(rng.standard_normal()))
        m += 1
    return cuadmm_amd.Problem(p.vec_len, m, p.blk, np.array(cp, np.int32), np.array(rows, np.int32), np.array(vals), np.array(b_idx, np.int32),
                              np.array(b_vals), p.C_idx, p.C_vals)


def _run(prob, iters, sw, monkeypatch, fuse, stop_tol=0.0, rows=True, solve=True):
    # every switch through the ABI (cuadmm_set_option); psd_wave4_min = 1: the one-wavefront kernels for 32 < n <= 64 whatever the count
    s = cuadmm_amd.SDPSolver(verbose=False, options={"fuse_solve": 1 if solve else 0, "psd_wave4_min": 1, "fuse": 1 if fuse else 0,
                                                     "fuse_rows": 1 if rows else 0})
    s.init_problem(prob)
    s.solve(iters, stop_tol, 0, 50, 100, sw, 1.05)
    return s


def _same(a, b, rtol=1e-9, atol=1e-12):
    a, b = np.asarray(a, float), np.asarray(b, float)
    assert a.shape == b.shape
    assert np.max(np.abs(a - b) / (atol / rtol + np.abs(b))) <= rtol


BLKS = {
    "all32": [32] * 300,
    "mixed_c4": [3, 6, 10, 15, 28, 45] * 60,
    "every_class": [1, 4, 8, 16, 17, 31, 32, 33, 40, 48, 49, 57, 63, 64, 70, 100] * 6,
    "only_mid": [45] * 40 + [64] * 10 + [50] * 10,
}


@pytest.mark.parametrize("name", sorted(BLKS))
@pytest.mark.parametrize("sw", [0, 8, 1000])          # ADMM only | sGS then the switch (snapshots) | sGS only
def test_fused_iteration_equals_standalone_kernels(name, sw, monkeypatch):
    _, prob = _problem(BLKS[name])
    iters = 25
    a = _run(prob, iters, sw, monkeypatch, fuse=True)
    b = _run(prob, iters, sw, monkeypatch, fuse=False)
    assert a.info_iter_num == b.info_iter_num == iters
    for nm in ("errRp", "errRd", "pobj", "dobj", "relgap"):
        _same(a.info_arr(nm), b.info_arr(nm))
    assert np.array_equal(a.info_arr("sig"), b.info_arr("sig"))
    for va, vb in ((a.X, b.X), (a.y, b.y), (a.S, b.S)):
        assert np.max(np.abs(va - vb)) <= 1e-9 * (1 + np.max(np.abs(vb)))


def test_fused_iteration_vs_oracle(monkeypatch):
    """The fused engine against the numpy oracle on a mixed problem (20 iterations, both phases)."""
    p, prob = _problem([3, 6, 10, 15, 28, 45, 33, 64] * 12, seed=11)
    iters, sw = 20, 10
    s = _run(prob, iters, sw, monkeypatch, fuse=True)
    op = orc.Problem(p.vec_len, p.con_num, p.blk, p.At_col_ptrs, p.At_row_ids, p.At_vals, p.b_idx, p.b_vals, p.C_idx, p.C_vals)
    o = orc.OracleSolver().init_problem(op)
    info = o.solve(iters, 0.0, 0, 50, 100, sw, 1.05)
    for nm in ("errRp", "errRd", "pobj", "dobj", "relgap"):
        _same(s.info_arr(nm), np.asarray(getattr(info, nm), float), rtol=1e-8, atol=1e-11)
    assert np.max(np.abs(s.X - o.X)) <= 1e-8 * (1 + np.max(np.abs(o.X)))


def test_fused_runs_are_bit_reproducible(monkeypatch):
    _, prob = _problem(BLKS["mixed_c4"], seed=3)
    a = _run(prob, 15, 0, monkeypatch, fuse=True)
    b = _run(prob, 15, 0, monkeypatch, fuse=True)
    for nm in ("errRp", "errRd", "pobj", "dobj"):
        assert np.array_equal(a.info_arr(nm), b.info_arr(nm))
    assert np.array_equal(a.X, b.X) and np.array_equal(a.S, b.S)


@pytest.mark.parametrize("sw", [0, 6, 1000])
def test_local_constraint_rows_inside_the_projection_kernel(sw, monkeypatch):
    """Rows of A local to one fused block are evaluated by that block's kernel (SignFuse::lc_*), the coupled ones by the
    stand-alone SpMV over a compact row list: same trajectories as with CUADMM_FUSE_ROWS=0 and as without any fusion."""
    prob = _coupled_problem([32] * 40 + [45] * 12 + [12] * 30 + [6] * 20 + [70] * 2, n_couple=25)
    iters = 20
    a = _run(prob, iters, sw, monkeypatch, fuse=True, rows=True)
    b = _run(prob, iters, sw, monkeypatch, fuse=True, rows=False)
    c = _run(prob, iters, sw, monkeypatch, fuse=False)
    for other in (b, c):
        for nm in ("errRp", "errRd", "pobj", "dobj", "relgap"):
            _same(a.info_arr(nm), other.info_arr(nm))
        assert np.array_equal(a.info_arr("sig"), other.info_arr("sig"))
        for va, vb in ((a.X, other.X), (a.y, other.y), (a.S, other.S)):
            assert np.max(np.abs(va - vb)) <= 1e-9 * (1 + np.max(np.abs(vb)))


@pytest.mark.parametrize("sw", [0, 7, 1000])
@pytest.mark.parametrize("cons", [1, 3, 8])
def test_closed_blocks_solve_for_their_own_multipliers(cons, sw, monkeypatch):
    """Block-diagonal problems with at most 8 constraints per block: the projection kernel solves the block's part of
    A A^T y = rhs itself (same elimination order and unfused arithmetic as forest_solve_kernel) and adds its rows' share of
    ||Rp||^2 and b^T y -- same trajectories as with the stand-alone solve / statistics kernels (CUADMM_FUSE_SOLVE=0)."""
    _, prob = _problem([32] * 50 + [45] * 10 + [12] * 20 + [20] * 20, cons_per_block=cons, seed=13)
    iters = 30
    a = _run(prob, iters, sw, monkeypatch, fuse=True, solve=True)
    b = _run(prob, iters, sw, monkeypatch, fuse=True, solve=False)
    for nm in ("errRp", "errRd", "pobj", "dobj", "relgap"):
        _same(a.info_arr(nm), b.info_arr(nm), rtol=1e-10)
    assert np.array_equal(a.info_arr("sig"), b.info_arr("sig"))
    for va, vb in ((a.X, b.X), (a.y, b.y), (a.S, b.S)):
        assert np.max(np.abs(va - vb)) <= 1e-11 * (1 + np.max(np.abs(vb)))
