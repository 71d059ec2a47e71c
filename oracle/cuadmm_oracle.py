"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (numpy / scipy) of the hot path of ComputationalRobotics/cuADMM:
the body of ``SDPSolver::init`` / ``SDPSolver::solve`` and the helpers it calls.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module, and only as the checker.

Parity status: PINNED.  ``tests/test_oracle_pinning.py`` checks this restatement
against the reference's own known answers:
  * unit KATs hard-coded in the reference tests (svec maps, sqrt(2) scaling,
    COO->CSC ordering, eigen spectra, permutation scatter, normA), and
  * the console logs shipped with the reference
    (examples/benchmarks/ros_2000/{cuADMM,sGS-cuADMM}.log,
     examples/benchmarks/PushT_N=10_MOMENT/*.log, examples/plato/logs/rose13.log ...),
    transcribed into tests/golden/ref_logs.json by tests/golden/make_golden.py.

Third-party arithmetic absent from /root/reference (cuSOLVER Xsyevd / syevjBatched,
cuBLAS, cuSPARSE, CHOLMOD; versions unpinned by the reference's CMakeLists.txt:42-44,70)
is restated by its published mathematical contract: a symmetric eigendecomposition
(LAPACK dsyevd through numpy, the same routine as the reference's own eig_cpu path,
include/cuadmm/eig_cpu.h:31-51) and an exact solve with (A A^T + eps I).

Every function cites the reference file:line it follows (paths relative to the
reference repository root).
"""
from __future__ import annotations

import math
import os
import re
import time
from dataclasses import dataclass, field

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla


# --------------------------------------------------------------------------------------
# constants  (include/cuadmm/kernels.h:173-181)
# --------------------------------------------------------------------------------------
def _sqrt_newton_raphson(x: float, curr: float, prev: float) -> float:
    """constexpr sqrtNewtonRaphson, include/cuadmm/kernels.h:173-178 (fixed point of Newton)."""
    while curr != prev:
        curr, prev = 0.5 * (curr + x / curr), curr
    return curr


SQRT2 = _sqrt_newton_raphson(2.0, 2.0, 0.0)      # kernels.h:180 -> 0x1.6a09e667f3bccp+0
SQRT2INV = 1.0 / SQRT2                           # kernels.h:181 -> 0x1.6a09e667f3bcdp-1


# --------------------------------------------------------------------------------------
# TXT input  (src/utils/io.cu, src/problem.cu)
# --------------------------------------------------------------------------------------
_BLK_TYPE_VAL = re.compile(r"^\s*([a-zA-Z])\s+(-?\d+)\s*$")
_BLK_VAL = re.compile(r"^\s*(-?\d+)\s*$")


def read_blk(path: str):
    """read_blk, src/utils/io.cu:296-329: 's 10' or '10' (bare int => 's'); malformed lines ignored."""
    out = []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            m = _BLK_TYPE_VAL.match(line)
            if m:
                out.append((m.group(1), int(m.group(2))))
                continue
            m = _BLK_VAL.match(line)
            if m:
                out.append(("s", int(m.group(1))))
    return out


def _read_numbers(path: str) -> np.ndarray:
    with open(path) as f:
        txt = f.read().split()
    return np.array([float(t) for t in txt], dtype=np.float64)


def read_coo(path: str):
    """read_COO_sparse_matrix_data, src/utils/io.cu:96-125: whitespace triplets 'row col val'."""
    a = _read_numbers(path)
    a = a[: (a.size // 3) * 3].reshape(-1, 3)
    return a[:, 0].astype(np.int32), a[:, 1].astype(np.int32), a[:, 2].copy()


def read_sparse_vector(path: str):
    """read_sparse_vector_data, src/utils/io.cu:68-93: triplets 'idx 0 val'."""
    r, _c, v = read_coo(path)
    return r, v


def coo_to_csc(col_ids, row_ids, vals, col_num):
    """COO_to_CSC, src/utils/io.cu:187-243: sort by (col,row) lexicographically, build col_ptrs.

    (The reference mis-builds col_ptrs when column 0 is empty; every
    shipped data set has column 0 non-empty, so the correct conversion below coincides.)
    """
    order = np.lexsort((row_ids, col_ids))
    col_ids = np.asarray(col_ids)[order]
    row_ids = np.asarray(row_ids)[order]
    vals = np.asarray(vals)[order]
    col_ptrs = np.zeros(col_num + 1, dtype=np.int32)
    np.add.at(col_ptrs, col_ids + 1, 1)
    col_ptrs = np.cumsum(col_ptrs).astype(np.int32)
    return col_ptrs, row_ids.astype(np.int32), vals.astype(np.float64)


@dataclass
class Problem:
    """Problem::from_txt, src/problem.cu:11-83."""
    vec_len: int
    con_num: int
    blk: np.ndarray                 # int32 block sizes, blk.txt order
    At_col_ptrs: np.ndarray         # CSC of At (vec_len x con_num)
    At_row_ids: np.ndarray
    At_vals: np.ndarray
    b_idx: np.ndarray
    b_vals: np.ndarray
    C_idx: np.ndarray
    C_vals: np.ndarray

    @property
    def At_nnz(self):
        return int(self.At_vals.size)


def load_problem_txt(prefix: str) -> Problem:
    if not prefix.endswith("/"):
        prefix += "/"           # main.cu concatenates; callers of the reference must pass the '/'
    blk_pairs = read_blk(prefix + "blk.txt")
    vec_len = 0
    for t, n in blk_pairs:
        if t != "s":
            raise ValueError(f"unknown block type '{t}' in blk.txt")   # problem.cu:28-36
        vec_len += n * (n + 1) // 2
    con_num = int(_read_numbers(prefix + "con_num.txt")[0])
    rows, cols, vals = read_coo(prefix + "At.txt")
    cp, ri, v = coo_to_csc(cols, rows, vals, con_num)
    b_idx, b_vals = read_sparse_vector(prefix + "b.txt")
    if os.path.getsize(prefix + "C.txt") > 0:
        C_idx, C_vals = read_sparse_vector(prefix + "C.txt")
    else:
        C_idx, C_vals = np.zeros(0, np.int32), np.zeros(0)
    return Problem(vec_len, con_num, np.array([n for _, n in blk_pairs], dtype=np.int32),
                   cp, ri, v, b_idx, b_vals, C_idx, C_vals)


# --------------------------------------------------------------------------------------
# block bookkeeping / svec maps
# --------------------------------------------------------------------------------------
def is_large_mat(mat_size: int, mat_num: int) -> bool:
    """is_large_mat, src/matrix_sizes.cu:14-19."""
    if mat_size > 32:
        return True
    return float(mat_size) - 17.0 > float(mat_num) * 1.4


def analyze_blk(blk):
    """analyze_blk, src/utils/analyze_blk.cu:63-99: ascending unique sizes + multiplicities."""
    sizes = sorted(set(int(x) for x in blk))
    nums = [int(np.sum(np.asarray(blk) == s)) for s in sizes]
    return sizes, nums


class MatrixSizes:
    """MatrixSizes::init and offset helpers, src/matrix_sizes.cu:22-69,116-151."""

    def __init__(self, blk_sizes, blk_nums):
        self.is_large_map = {}
        self.large_mat_sizes, self.large_mat_nums = [], []
        self.small_mat_sizes, self.small_mat_nums = [], []
        self.large_mat_start_indices, self.large_W_start_indices = [0], [0]
        self.small_mat_start_indices, self.small_W_start_indices = [0], [0]
        self.total_large_mat_size = self.total_small_mat_size = 0
        self.sum_large_mat_size = self.sum_small_mat_size = 0
        self.large_mat_num = self.small_mat_num = 0
        for s, c in zip(blk_sizes, blk_nums):
            big = is_large_mat(s, c)
            self.is_large_map[s] = big
            if big:
                self.large_mat_num += c
                self.sum_large_mat_size += s * c
                self.total_large_mat_size += c * s * s
                self.large_mat_sizes.append(s)
                self.large_mat_nums.append(c)
                self.large_mat_start_indices.append(self.total_large_mat_size)
                self.large_W_start_indices.append(self.sum_large_mat_size)
            else:
                self.small_mat_num += c
                self.sum_small_mat_size += s * c
                self.total_small_mat_size += c * s * s
                self.small_mat_sizes.append(s)
                self.small_mat_nums.append(c)
                self.small_mat_start_indices.append(self.total_small_mat_size)
                self.small_W_start_indices.append(self.sum_small_mat_size)

    def is_large(self, s):
        return self.is_large_map[s]

    def large_mat_offset(self, idx, same):
        return self.large_mat_start_indices[idx] + same * self.large_mat_sizes[idx] ** 2

    def small_mat_offset(self, idx, same=0):
        return self.small_mat_start_indices[idx] + same * self.small_mat_sizes[idx] ** 2


def get_maps(blk, sizes: MatrixSizes):
    """get_maps, src/utils/get_maps.cu:80-135.  Returns int32 (map_B, map_M1, map_M2)."""
    vec_len = int(sum(int(s) * (int(s) + 1) // 2 for s in blk))
    map_B = np.empty(vec_len, np.int32)
    map_M1 = np.empty(vec_len, np.int32)
    map_M2 = np.empty(vec_len, np.int32)
    seen_large = [0] * len(sizes.large_mat_sizes)
    seen_small = [0] * len(sizes.small_mat_sizes)
    idx = 0
    for s in blk:
        s = int(s)
        if sizes.is_large(s):
            b = 0
            k = sizes.large_mat_sizes.index(s)
            base = sizes.large_mat_offset(k, seen_large[k])
            seen_large[k] += 1
        else:
            b = 1
            k = sizes.small_mat_sizes.index(s)
            base = sizes.small_mat_offset(k, seen_small[k])
            seen_small[k] += 1
        ii, jj = np.tril_indices(s)          # i = 1..s outer, j = 1..i inner (0-based here)
        cnt = ii.size
        map_B[idx:idx + cnt] = b
        map_M1[idx:idx + cnt] = base + s * ii + jj     # get_maps.cu:121,127
        map_M2[idx:idx + cnt] = base + s * jj + ii     # get_maps.cu:123,129
        idx += cnt
    return map_B, map_M1, map_M2


def get_maps_duo(blk, LARGE, SMALL):
    """get_maps_duo, src/utils/get_maps.cu:21-68."""
    vec_len = int(sum(int(s) * (int(s) + 1) // 2 for s in blk))
    map_B = np.empty(vec_len, np.int32)
    map_M1 = np.empty(vec_len, np.int32)
    map_M2 = np.empty(vec_len, np.int32)
    k_mom = k_loc = 0
    idx = 0
    for s in blk:
        s = int(s)
        if s == LARGE:
            b = 0
            k_mom += 1
            base = s * s * (k_mom - 1)
        else:
            b = 1
            k_loc += 1
            base = s * s * (k_loc - 1)
        ii, jj = np.tril_indices(s)
        cnt = ii.size
        map_B[idx:idx + cnt] = b
        map_M1[idx:idx + cnt] = base + s * ii + jj
        map_M2[idx:idx + cnt] = base + s * jj + ii
        idx += cnt
    return map_B, map_M1, map_M2


def vector_to_matrices(Xb, large_mat, small_mat, map_B, map_M1, map_M2):
    """vector_to_matrices_kernel, src/kernels/vec_mat_conversion.cu:11-34."""
    diag = (map_M1 == map_M2)
    v = Xb * (SQRT2INV + diag.astype(np.float64) * (1 - SQRT2INV))
    lg = map_B == 0
    large_mat[map_M1[lg]] = v[lg]
    large_mat[map_M2[lg]] = v[lg]
    sm = ~lg
    small_mat[map_M1[sm]] = v[sm]
    small_mat[map_M2[sm]] = v[sm]


def matrices_to_vector(large_mat, small_mat, map_B, map_M1, map_M2):
    """matrices_to_vector_kernel, src/kernels/vec_mat_conversion.cu:36-57 (reads M1 only)."""
    diag = (map_M1 == map_M2)
    f = (SQRT2 + diag.astype(np.float64) * (1 - SQRT2))
    out = np.empty(map_B.size)
    lg = map_B == 0
    out[lg] = large_mat[map_M1[lg]] * f[lg]
    out[~lg] = small_mat[map_M1[~lg]] * f[~lg]
    return out


def blk_svec_len(blk):
    """svec slots per block: n(n+1)/2 for a PSD block 's n'; an unconstrained block 'u n' (README.md:55-64, WIP in the
    reference, whose loader still rejects it, problem.cu:28-36) is carried as the NEGATIVE size -n and owns n slots."""
    blk = np.asarray(blk, dtype=np.int64)
    return np.where(blk >= 0, blk * (blk + 1) // 2, -blk)


def svec_block_offsets(blk):
    """svec offsets of each block in blk.txt order (get_maps.cu:116-117 walk)."""
    blk = np.asarray(blk, dtype=np.int64)
    off = np.zeros(blk.size + 1, dtype=np.int64)
    np.cumsum(blk_svec_len(blk), out=off[1:])
    return off


class BlockIndex:
    """Groups blocks of equal size so svec<->dense conversion and eigh can be batched."""

    def __init__(self, blk):
        self.blk = np.asarray(blk, dtype=np.int64)
        self.off = svec_block_offsets(self.blk)
        self.groups = []
        # unconstrained blocks (negative size): their svec ranges pass through the projection unchanged
        self.free = [(int(self.off[k]), int(self.off[k + 1])) for k in np.nonzero(self.blk < 0)[0]]
        for n in sorted(set(int(x) for x in self.blk if x > 0)):
            ids = np.nonzero(self.blk == n)[0]
            ii, jj = np.tril_indices(n)       # svec slot t <-> (col ii[t], row jj[t]), jj<=ii
            gather = self.off[ids][:, None] + np.arange(n * (n + 1) // 2)[None, :]
            self.groups.append((n, ids, ii, jj, gather))

    def unpack(self, x):
        """svec -> list of (n, dense (cnt,n,n)) with 1/sqrt2 off-diagonals (vec_mat_conversion.cu:26)."""
        out = []
        for n, ids, ii, jj, gather in self.groups:
            seg = x[gather]
            scale = np.where(ii == jj, 1.0, SQRT2INV)
            M = np.zeros((ids.size, n, n))
            v = seg * scale[None, :]
            M[:, jj, ii] = v
            M[:, ii, jj] = v
            out.append(M)
        return out

    def pack(self, mats):
        """dense -> svec with sqrt2 off-diagonals, reading the upper element (vec_mat_conversion.cu:51)."""
        x = np.zeros(int(self.off[-1]))
        for (n, ids, ii, jj, gather), M in zip(self.groups, mats):
            scale = np.where(ii == jj, 1.0, SQRT2)
            x[gather] = M[:, jj, ii] * scale[None, :]
        return x


def psd_project_svec(bidx: BlockIndex, xb: np.ndarray, return_eigs=False, eig_rank=0):
    """Steps solver.cu:534-647: unpack, eig, max(W,0), V diag(W) V^T, pack.

    The eigendecomposition is LAPACK dsyevd (numpy.linalg.eigh), i.e. the reference's
    eig_cpu routine (eig_cpu.h:31-51); the GPU reference uses cuSOLVER (cusolver.h:86,164).
    eig_rank > 0: the rank-limited projection the reference prepares but leaves switched off
    (duo_solver.cu:428-438,843-850): W = max(W,0) * mask (dense_scalar.cu:51-57) with mask = 1 on the LAST eig_rank
    entries of the ascending eigenvalues of every block (get_eig_rank_mask.cu:13-37), i.e. only the eig_rank largest
    eigenvalues survive.  Unconstrained blocks (negative size) are copied through.
    """
    mats = bidx.unpack(xb)
    proj = []
    eigs = []
    for M in mats:
        w, V = np.linalg.eigh(M)
        wp = np.maximum(w, 0.0)                                  # dense_scalar.cu:41-47
        if eig_rank > 0 and eig_rank < wp.shape[1]:
            wp[:, : wp.shape[1] - eig_rank] = 0.0                # get_eig_rank_mask.cu:30-35 (ascending: last r kept)
        tmp = V * wp[:, None, :]                                 # diagonal_batch.cu:11-23
        proj.append(tmp @ np.swapaxes(V, 1, 2))                  # cublas.h:18-35 (N,T)
        eigs.append(w)
    x = bidx.pack(proj)
    for lo, hi in bidx.free:
        x[lo:hi] = xb[lo:hi]
    return (x, eigs) if return_eigs else x


# --------------------------------------------------------------------------------------
# the solver
# --------------------------------------------------------------------------------------
LOG_ROW_FMT = " %4d | %3.2e %3.2e | %- 5.4e %- 5.4e %3.2e | %5.1f | %2.1e |"     # solver.cu:440


@dataclass
class SolveInfo:
    iter_num: int = 0
    pobj: list = field(default_factory=list)
    dobj: list = field(default_factory=list)
    errRp: list = field(default_factory=list)
    errRd: list = field(default_factory=list)
    relgap: list = field(default_factory=list)
    sig: list = field(default_factory=list)
    log_rows: list = field(default_factory=list)      # (iter-1, errRp, errRd, pobj, dobj, relgap, sig)
    final_msg: str = ""


class OracleSolver:
    """Restatement of SDPSolver (include/cuadmm/solver.h:30-248, src/solver.cu)."""

    def __init__(self, eig_fn=None, eig_rank=0, eig_rank_begin_iter=0, eig_rank_maxfeas=0.0):
        self.eig_fn = eig_fn
        # rank-limited projection, dormant in the reference (duo_solver.cu:428 eig_rank = 5; :843-850 the switch
        # `iter >= begin_low_rank_proj || maxfeas < 1e-3` is commented out): 0 = off
        self.eig_rank, self.eig_rank_begin_iter, self.eig_rank_maxfeas = int(eig_rank), int(eig_rank_begin_iter), float(eig_rank_maxfeas)

    # -- SDPSolver::init, src/solver.cu:27-342 -----------------------------------------
    def init(self, vec_len, con_num, At_col_ptrs, At_row_ids, At_vals,
             b_idx, b_vals, C_idx, C_vals, blk, X=None, y=None, S=None, sig=1.0):
        self.vec_len, self.con_num = int(vec_len), int(con_num)
        L, m = self.vec_len, self.con_num
        At_vals = np.array(At_vals, dtype=np.float64)
        cp = np.asarray(At_col_ptrs, dtype=np.int64)

        # get_normA, src/kernels/sparse_matrix_norm.cu:11-31  (columns of At = constraints)
        sq = At_vals * At_vals
        colsum = np.add.reduceat(np.append(sq, 0.0), np.minimum(cp[:-1], sq.size))
        colsum[cp[:-1] == cp[1:]] = 0.0
        self.normA = np.maximum(1.0, np.sqrt(colsum))
        cols = np.repeat(np.arange(m), np.diff(cp))
        At_vals = At_vals / self.normA[cols]

        self.At = sp.csc_matrix((At_vals, np.asarray(At_row_ids), np.asarray(At_col_ptrs)), shape=(L, m))
        self.A = self.At.T.tocsr()            # solver.cu:86-88: same arrays viewed as CSR of A
        self.At_csr = self.At.tocsr()         # solver.cu:83-85

        # CholeskySolverCPU::get_A/factorize, cholesky_cpu.h:62-141, eps=1e-15 (solver.cu:94)
        AAt = (self.A @ self.At).tocsc() + 1e-15 * sp.identity(m, format="csc")
        self._solve = spla.factorized(AAt)

        self.b = np.zeros(m); self.b[np.asarray(b_idx, dtype=np.int64)] = np.asarray(b_vals, dtype=np.float64)
        self.C = np.zeros(L); self.C[np.asarray(C_idx, dtype=np.int64)] = np.asarray(C_vals, dtype=np.float64)
        self.b_idx = np.asarray(b_idx, dtype=np.int64)
        self.C_idx = np.asarray(C_idx, dtype=np.int64)
        self.X = np.zeros(L) if X is None else np.array(X, dtype=np.float64)
        self.y = np.zeros(m) if y is None else np.array(y, dtype=np.float64)
        self.S = np.zeros(L) if S is None else np.array(S, dtype=np.float64)
        self.sig = float(sig)

        self.blk = np.asarray(blk, dtype=np.int64)
        self.bidx = BlockIndex(self.blk)

        # scaling, solver.cu:169-191
        self.norm_borg = 1 + float(np.linalg.norm(np.asarray(b_vals, dtype=np.float64)))
        self.norm_Corg = 1 + float(np.linalg.norm(np.asarray(C_vals, dtype=np.float64)))
        self.b[self.b_idx] = self.b[self.b_idx] / self.normA[self.b_idx]      # :181
        self.y = self.y * self.normA                                          # :182
        self.bscale = 1 + float(np.linalg.norm(self.b[self.b_idx]))           # :184
        self.Cscale = 1 + float(np.linalg.norm(self.C[self.C_idx]))           # :185
        self.objscale = self.bscale * self.Cscale
        self.b *= (1 / self.bscale)                                           # sparse_scalar.cu:41-54
        self.C *= (1 / self.Cscale)
        self.X *= (1 / self.bscale)                                           # dense_scalar.cu:77-81
        self.S *= (1 / self.Cscale)
        self.y *= (1 / self.Cscale)

        # initial residuals, solver.cu:195-228
        self.Aty = self.At_csr @ self.y
        self.Rp = -(self.A @ self.X) + self.b
        self.SmC = self.S - self.C
        self.Rd = self.Aty + self.SmC
        self.errRp = float(np.linalg.norm(self.normA * self.Rp * self.bscale)) / self.norm_borg
        self.errRd = float(np.linalg.norm(self.Rd * self.Cscale)) / self.norm_Corg
        self.maxfeas = max(self.errRp, self.errRd)
        self.pobj = float(self.C @ self.X) * self.objscale
        self.dobj = float(self.b @ self.y) * self.objscale
        self.relgap = abs(self.pobj - self.dobj) / (1 + abs(self.pobj) + abs(self.dobj))

        # others, solver.cu:323-339
        self.prim_win = 0
        self.dual_win = 0
        self.ratioconst = 1e0
        self.sigmax = 1e3
        self.sigmin = 1e-3
        self.best_KKT = 0.0
        self.info = SolveInfo()
        self.t0 = time.time()
        return self

    def init_problem(self, p: Problem, X=None, y=None, S=None, sig=1.0):
        return self.init(p.vec_len, p.con_num, p.At_col_ptrs, p.At_row_ids, p.At_vals,
                         p.b_idx, p.b_vals, p.C_idx, p.C_vals, p.blk, X, y, S, sig)

    def _linsys(self, rhs):
        # perform_permutation + cholmod_solve2(LDLt) + perform_permutation, solver.cu:487-500.
        # P^T (L D L^T)^-1 P applied to rhs == (A A^T + eps I)^-1 rhs.
        return self._solve(rhs)

    def project(self, xb, it=0):
        if self.eig_fn is not None:
            return self.eig_fn(self.bidx, xb)
        rank = 0
        if self.eig_rank > 0 and (it >= self.eig_rank_begin_iter or self.maxfeas < self.eig_rank_maxfeas):   # duo_solver.cu:844
            rank = self.eig_rank
        return psd_project_svec(self.bidx, xb, eig_rank=rank)

    # -- SDPSolver::solve, src/solver.cu:355-823 ------------------------------------------
    def solve(self, max_iter, stop_tol, sig_update_threshold=500, sig_update_stage_1=50,
              sig_update_stage_2=100, switch_admm=11000, sigscale=1.05, if_first=True,
              verbose=False, stage_hook=None):
        sig_update_threshold = int(sig_update_threshold)
        info = self.info
        info.iter_num = 0
        breakyes = False
        A, At_csr = self.A, self.At_csr

        if not if_first:                                                      # solver.cu:385-409
            self.y = self.y * self.normA
            self.X = self.X * (1 / self.bscale)
            self.S = self.S * (1 / self.Cscale)
            self.y = self.y * (1 / self.Cscale)
            self.SmC = self.S - self.C
            self.Rp = -(A @ self.X) + self.b

        X_best = y_best = S_best = None
        it = 1
        while it <= max_iter + 1:
            # Step 0, solver.cu:419-467
            if max(self.maxfeas, self.relgap) < stop_tol:
                breakyes = True
                info.final_msg = "Solver ended: converged."
            if it > max_iter:
                breakyes = True
                info.final_msg = "Solver ended: maximum iteration reached"
            if breakyes or (it <= 200 and it % 50 == 1) or (it > 200 and it % 100 == 1):
                row = (it - 1, self.errRp, self.errRd, self.pobj, self.dobj, self.relgap, self.sig)
                info.log_rows.append(row)
                if verbose:
                    print(LOG_ROW_FMT % (row[0], row[1], row[2], row[3], row[4], row[5],
                                         time.time() - self.t0, row[6]), flush=True)

            # Step 1, solver.cu:478-500
            rhsy = -(A @ self.SmC)
            rhsy += (1 / self.sig) * self.Rp
            self.y = self._linsys(rhsy)

            # Step 2, solver.cu:514-527
            self.Aty = At_csr @ self.y
            Rd1 = self.Aty - self.C
            Xb = self.X + Rd1 * self.sig
            if breakyes:                                                      # solver.cu:567-576
                if it > switch_admm and X_best is not None:
                    self.X, self.y, self.S = X_best.copy(), y_best.copy(), S_best.copy()
                break
            Xproj = self.project(Xb, it)                                      # solver.cu:534-647
            Xdiff = 1.0 * Xproj + (-1.0) * self.X                             # :652
            self.S = (1 / self.sig) * Xdiff + (-1.0) * Rd1                    # :656
            if stage_hook is not None:
                stage_hook(it, "proj", Xb=Xb, Xproj=Xproj, S=self.S)

            # Step 3, solver.cu:672-741
            self.SmC = self.S - self.C
            if it == switch_admm:
                sig_update_stage_2 = sig_update_stage_2 // 2
                sigscale = sigscale * 1.23
                self.best_KKT = max(self.maxfeas, self.relgap)
                X_best, y_best, S_best = self.X.copy(), self.y.copy(), self.S.copy()
            if it < switch_admm:
                rhsy = -(A @ self.SmC)
                rhsy += (1 / self.sig) * self.Rp
                self.y = self._linsys(rhsy)
                self.Aty = At_csr @ self.y
                Rd1 = self.Aty - self.C
            if it > switch_admm:
                if X_best is not None and self.best_KKT > max(self.maxfeas, self.relgap):
                    X_best, y_best, S_best = self.X.copy(), self.y.copy(), self.S.copy()
                    self.best_KKT = max(self.maxfeas, self.relgap)

            # Step 4, solver.cu:746-758
            self.Rd = 1.0 * Rd1 + 1.0 * self.S
            tau = 1.95 if it < switch_admm else 1.618
            if self.errRd < stop_tol:
                tau = max(1.618, tau / 1.1)
            self.X = 1.0 * self.X + (tau * self.sig) * self.Rd

            # Step 5, solver.cu:764-799
            self.Rp = -(A @ self.X) + self.b
            self.errRp = float(np.linalg.norm(self.normA * self.Rp * self.bscale)) / self.norm_borg
            self.pobj = float(self.C @ self.X) * self.objscale
            self.errRd = float(np.linalg.norm(self.Rd * self.Cscale)) / self.norm_Corg
            self.dobj = float(self.b @ self.y) * self.objscale
            self.maxfeas = max(self.errRp, self.errRd)
            self.relgap = abs(self.pobj - self.dobj) / (1 + abs(self.pobj) + abs(self.dobj))
            feasratio = self.ratioconst * self.errRp / self.errRd
            if feasratio < 1:
                self.prim_win += 1
            else:
                self.dual_win += 1
            if ((it <= sig_update_threshold and it % sig_update_stage_1 == 1) or
                    (it > sig_update_threshold and it % sig_update_stage_2 == 1)):
                if self.prim_win > 1.2 * self.dual_win:
                    self.prim_win = 0
                    self.sig = min(self.sigmax, self.sig * sigscale)
                elif self.dual_win > 1.2 * self.prim_win:
                    self.dual_win = 0
                    self.sig = max(self.sigmin, self.sig / sigscale)

            info.pobj.append(self.pobj); info.dobj.append(self.dobj)
            info.errRp.append(self.errRp); info.errRd.append(self.errRd)
            info.relgap.append(self.relgap); info.sig.append(self.sig)
            info.iter_num += 1
            if stage_hook is not None:
                stage_hook(it, "end", X=self.X, y=self.y, S=self.S)
            it += 1

        # unscale, solver.cu:814-816
        self.X = self.X * self.bscale
        self.y = self.y / self.normA * self.Cscale
        self.S = self.S * self.Cscale
        return info


def format_log_row(row, seconds=0.0):
    return LOG_ROW_FMT % (row[0], row[1], row[2], row[3], row[4], row[5], seconds, row[6])
