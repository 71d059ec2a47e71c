"""Seeded synthetic SDP instances for the benchmark configurations.

C2: blk = [32]*10000, 5 constraints per block each touching 8 svec slots of one block, strictly
feasible primal/dual pair by construction, dense C.  C4: mixed sizes {3,6,10,15,28,45}.
Pure numpy; no device code and no dependency on the test oracle.
"""
import numpy as np

SQRT2 = float.fromhex("0x1.6a09e667f3bccp+0")      # the reference's constant (include/cuadmm/kernels.h:180)


class SyntheticProblem:
    def __init__(self, vec_len, con_num, blk, At_col_ptrs, At_row_ids, At_vals, b_idx, b_vals, C_idx, C_vals):
        self.vec_len, self.con_num, self.blk = int(vec_len), int(con_num), np.asarray(blk, np.int32)
        self.At_col_ptrs, self.At_row_ids, self.At_vals = At_col_ptrs, At_row_ids, At_vals
        self.b_idx, self.b_vals, self.C_idx, self.C_vals = b_idx, b_vals, C_idx, C_vals

    @property
    def At_nnz(self):
        return int(self.At_vals.size)


def _offsets(blk):
    blk = np.asarray(blk, np.int64)
    off = np.zeros(blk.size + 1, np.int64)
    np.cumsum(blk * (blk + 1) // 2, out=off[1:])
    return off


def _pack(blk, off, groups, mats):
    x = np.empty(int(off[-1]))
    for (n, ids), M in zip(groups, mats):
        ii, jj = np.tril_indices(n)                      # slot t <-> (col ii[t], row jj[t])
        scale = np.where(ii == jj, 1.0, SQRT2)
        gather = off[ids][:, None] + np.arange(n * (n + 1) // 2)[None, :]
        x[gather] = M[:, jj, ii] * scale[None, :]
    return x


def make_synthetic(blk, cons_per_block=5, nnz_per_con=8, seed=20240601, dense_C=True):
    """Draw order (numpy PCG64(seed)): per size group (ascending n): G then H ~ N(0,1)^(cnt,n,n);
    per (r, segment length) constraint group: slot uniforms, then values ~ N(0,1); finally y0 ~ N(0,1)^m.
    X0 = G G^T/n + I, S0 = H H^T/n + I, b = A svec(X0), C = svec(S0) + A^T y0."""
    rng = np.random.Generator(np.random.PCG64(seed))
    blk = np.asarray(blk, dtype=np.int64)
    off = _offsets(blk)
    nb, L = blk.size, int(off[-1])
    groups = [(int(n), np.nonzero(blk == n)[0]) for n in sorted(set(int(x) for x in blk))]
    X0m, S0m = [], []
    for n, ids in groups:
        G = rng.standard_normal((ids.size, n, n))
        H = rng.standard_normal((ids.size, n, n))
        X0m.append(G @ np.swapaxes(G, 1, 2) / n + np.eye(n)[None])
        S0m.append(H @ np.swapaxes(H, 1, 2) / n + np.eye(n)[None])
    x0 = _pack(blk, off, groups, X0m)
    s0 = _pack(blk, off, groups, S0m)
    m = cons_per_block * nb
    seglen = blk * (blk + 1) // 2
    con_blk = np.arange(m) % nb
    r_eff = np.minimum(nnz_per_con, seglen[con_blk])
    rows_l, cols_l, vals_l = [], [], []
    for r in sorted(set(int(x) for x in r_eff)):
        cons = np.nonzero(r_eff == r)[0]
        for sl in sorted(set(int(x) for x in seglen[con_blk[cons]])):
            cc = cons[seglen[con_blk[cons]] == sl]
            u = rng.random((cc.size, sl))
            slots = np.argsort(u, axis=1)[:, :r]             # r distinct slots per constraint
            rows_l.append((off[con_blk[cc]][:, None] + slots).ravel())
            cols_l.append(np.repeat(cc, r))
            vals_l.append(rng.standard_normal(cc.size * r))
    rows = np.concatenate(rows_l)
    cols = np.concatenate(cols_l)
    vals = np.concatenate(vals_l)
    order = np.lexsort((rows, cols))                          # CSC of At: by constraint, then svec row
    rows, cols, vals = rows[order].astype(np.int32), cols[order].astype(np.int32), vals[order]
    cp = np.zeros(m + 1, np.int32)
    np.add.at(cp, cols + 1, 1)
    cp = np.cumsum(cp).astype(np.int32)
    y0 = rng.standard_normal(m)
    b = np.zeros(m)
    np.add.at(b, cols, vals * x0[rows])                       # b = A svec(X0)
    aty = np.zeros(L)
    np.add.at(aty, rows, vals * y0[cols])                     # A^T y0
    if dense_C:
        Cv = s0 + aty
    else:
        eye = _pack(blk, off, groups, [np.broadcast_to(np.eye(n), (ids.size, n, n)) for n, ids in groups])
        Cv = aty + eye
    b_idx = np.nonzero(b)[0].astype(np.int32)
    C_idx = np.nonzero(Cv)[0].astype(np.int32)
    return SyntheticProblem(L, m, blk.astype(np.int32), cp, rows, vals, b_idx, b[b_idx], C_idx, Cv[C_idx])


def config_c2(n_blocks=10000, n=32, seed=20240601):
    """BASELINE config 2: 10 000 PSD blocks, all 32x32."""
    return make_synthetic([n] * n_blocks, seed=seed)


def config_c4_blk(n_blocks=100000, seed=20240601):
    sizes = np.array([3, 6, 10, 15, 28, 45])
    blk = np.repeat(sizes, n_blocks // 6 + 1)[:n_blocks]
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    return blk[rng.permutation(blk.size)]


def config_c4(n_blocks=100000, seed=20240601):
    """BASELINE config 4: sizes {3,6,10,15,28,45} in equal shares, shuffled with the seed, 3 constraints/block."""
    return make_synthetic(config_c4_blk(n_blocks, seed), cons_per_block=3, seed=seed)


def config_c3(n=2000, p=0.01, seed=20240601):
    """BASELINE config 3 as SURVEY.md section 8d specifies it: max-cut relaxation of an Erdos-Renyi graph G(n, p), p = 0.01
    (about 20 neighbours per node at n = 2000), unit weights, one PSD block of size n (examples/max-cut/genMAXCUT.m:28-31).

        min <C, X>  s.t.  X_ii = 1 (i = 1..n),  X >= 0,     C = -(Diag(W 1) - W) / 4

    W: symmetric 0/1 adjacency, every pair i < j an edge with probability p (one uniform draw per pair, pairs in row-major order
    of the strict upper triangle).  m = n constraints, each a single svec slot.
    """
    rng = np.random.Generator(np.random.PCG64(seed + 2))
    lo, hi = np.triu_indices(n, 1)
    keep = rng.random(lo.size) < p
    lo, hi = lo[keep].astype(np.int64), hi[keep].astype(np.int64)
    deg = np.bincount(lo, minlength=n) + np.bincount(hi, minlength=n)
    L = n * (n + 1) // 2
    diag_slot = (np.arange(n, dtype=np.int64) * (np.arange(n, dtype=np.int64) + 1)) // 2 + np.arange(n)   # (col i, row i)
    off_slot = hi.astype(np.int64) * (hi + 1) // 2 + lo                                              # (col hi, row lo)
    C = np.zeros(L)
    C[diag_slot] = -deg / 4.0
    C[off_slot] = SQRT2 * (1.0 / 4.0)            # svec scaling of off-diagonal entries; -(-W_ij)/4
    cp = np.arange(n + 1, dtype=np.int32)
    rows = diag_slot.astype(np.int32)
    vals = np.ones(n)
    b_idx = np.arange(n, dtype=np.int32)
    C_idx = np.nonzero(C)[0].astype(np.int32)
    return SyntheticProblem(L, n, np.array([n], np.int32), cp, rows, vals, b_idx, np.ones(n), C_idx, C[C_idx])
