"""Input-side converters: the data formats the reference's examples ship in -> `cuadmm_amd.Problem` / the TXT directory
that `cuadmm_exe` reads.

The reference does these conversions in MATLAB (examples/sedumi_to_txt.m, sdpa_to_txt.m, mosek_to_txt.m with
examples/utils/read_sedumi.m, read_sdpa.m, convert_mosek2sedumi.m, svecADMM.m); a user of the engine has no MATLAB
in the loop, so the same mappings are provided here in numpy/scipy:

    problem_from_sedumi(A, b, c, K)   SeDuMi  min c'x s.t. Ax = b, x in K   (PSD blocks K.s, column-major vec)
    problem_from_sedumi_mat(path)     .mat with A (or At), b, c, K
    problem_from_svec_mat(path, blk)  .mat with At, C, b already in svec form (examples/plato/MATLAB/*.mat)
    problem_from_mosek_mat(path)      .mat with a MOSEK `prob` struct (bardim, bara, barc, blc)
    problem_from_sdpa(path)           SDPA sparse format .dat-s (semidefinite blocks)
    write_txt(problem, directory)     blk.txt, con_num.txt, At.txt, b.txt, C.txt (0-based 'row col val' triplets)

svec convention (include/cuadmm/kernels.h:180-181, vec_mat_conversion.cu): upper triangle column by column,
off-diagonal entries times sqrt(2).  Only semidefinite ('s') blocks exist in the engine (problem.cu:28-36): free,
linear, quadratic-cone or diagonal blocks are rejected with an error rather than silently reinterpreted.
"""
import os

import numpy as np
import scipy.io as sio
import scipy.sparse as sp

from .solver import Problem

SQRT2 = float.fromhex("0x1.6a09e667f3bccp+0")


def _svec_offsets(blk):
    blk = np.asarray(blk, dtype=np.int64)
    return np.concatenate([[0], np.cumsum(blk * (blk + 1) // 2)])


def _finish(blk, At, b, C):
    """At: (L x m) sparse, b: dense m, C: (L,) dense or sparse column -> Problem with sorted CSC arrays."""
    At = sp.csc_matrix(At)
    At.sum_duplicates()
    At.sort_indices()
    At.eliminate_zeros()
    b = np.asarray(b, dtype=np.float64).ravel()
    Cv = np.asarray(C.todense()).ravel() if sp.issparse(C) else np.asarray(C, dtype=np.float64).ravel()
    b_idx = np.nonzero(b)[0].astype(np.int32)
    C_idx = np.nonzero(Cv)[0].astype(np.int32)
    return Problem(int(At.shape[0]), int(At.shape[1]), np.asarray(blk, np.int32), At.indptr.astype(np.int32), At.indices.astype(np.int32),
                   At.data.astype(np.float64), b_idx, b[b_idx], C_idx, Cv[C_idx])


def _vec_to_svec_map(blk):
    """Sparse W (L x sum n^2) with svec(M) = W vec((M + M')/2) per block, vec column-major."""
    blk = np.asarray(blk, dtype=np.int64)
    off = _svec_offsets(blk)
    rows, cols, vals = [], [], []
    col0 = 0
    for k, n in enumerate(blk):
        r, c = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")        # entry (r, c) at vec position c*n + r
        r, c = r.ravel(order="F"), c.ravel(order="F")
        lo, hi = np.minimum(r, c), np.maximum(r, c)
        rows.append(off[k] + hi * (hi + 1) // 2 + lo)
        cols.append(col0 + c * n + r)
        vals.append(np.where(r == c, 1.0, SQRT2 / 2.0))                      # sqrt2 * (M_rc + M_cr) / 2
        col0 += n * n
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(int(off[-1]), int(col0)))


def _k_field(K, name):
    if isinstance(K, dict):
        v = K.get(name, [])
    elif hasattr(K, "dtype") and K.dtype.names:                              # scipy.io struct array
        v = K[name][0, 0] if name in K.dtype.names else []
    else:
        v = getattr(K, name, [])
    return np.atleast_1d(np.asarray(v)).astype(np.int64).ravel()


def problem_from_sedumi(A, b, c, K):
    """SeDuMi data (A is m x N or its transpose N x m; x = [PSD blocks as column-major vec]) -> Problem."""
    s = _k_field(K, "s")
    for other in ("f", "l", "q", "r"):
        v = _k_field(K, other)
        if v.size and int(v.sum()) != 0:
            raise ValueError("SeDuMi cone K.%s is not supported by the engine (semidefinite blocks only)" % other)
    if s.size == 0 or np.any(s < 1):
        raise ValueError("K.s must list the sizes of the semidefinite blocks")
    N = int(np.sum(s * s))
    b = np.asarray(b.todense() if sp.issparse(b) else b, dtype=np.float64).ravel()
    A = sp.csr_matrix(A)
    if A.shape == (N, b.size) and A.shape != (b.size, N):
        A = A.T.tocsr()
    if A.shape != (b.size, N):
        raise ValueError("A has shape %s, expected (%d, %d) for K.s" % (A.shape, b.size, N))
    W = _vec_to_svec_map(s)
    At = W @ A.T
    cv = np.asarray(c.todense() if sp.issparse(c) else c, dtype=np.float64).ravel()
    if cv.size != N:
        raise ValueError("c has %d entries, expected %d" % (cv.size, N))
    return _finish(s, At, b, W @ cv)


def problem_from_sedumi_mat(path):
    d = sio.loadmat(path)
    A = d["A"] if "A" in d else d["At"].T
    c = d["c"] if "c" in d else d["C"]
    return problem_from_sedumi(A, d["b"], c, d["K"])


def problem_from_svec_mat(path, blk):
    """.mat with At (L x m), C (L x 1), b (m) already in svec form (examples/plato/MATLAB/{1dc.1024,swissroll,...}.mat)."""
    d = sio.loadmat(path)
    L = int(_svec_offsets(blk)[-1])
    At = sp.csc_matrix(d["At"])
    if At.shape[0] != L:
        raise ValueError("At has %d rows, blk implies %d" % (At.shape[0], L))
    return _finish(blk, At, d["b"].todense() if sp.issparse(d["b"]) else d["b"], sp.csc_matrix(d["C"]))


def problem_from_mosek_mat(path):
    """MOSEK `prob` struct: bardim, bara.(subi,subj,subk,subl,val), barc.(subj,subk,subl,val), blc (= buc) -> Problem.
    MOSEK stores the lower triangle (k >= l); the svec slot of (k, l) is column k, row l of the upper triangle."""
    prob = sio.loadmat(path, squeeze_me=True, struct_as_record=False)["prob"]
    bardim = np.atleast_1d(prob.bardim).astype(np.int64)
    off = _svec_offsets(bardim)

    def slots(sub):
        j = np.atleast_1d(sub.subj).astype(np.int64) - 1
        k = np.atleast_1d(sub.subk).astype(np.int64) - 1
        l = np.atleast_1d(sub.subl).astype(np.int64) - 1
        v = np.atleast_1d(sub.val).astype(np.float64)
        return off[j] + k * (k + 1) // 2 + l, np.where(k != l, v * np.sqrt(2.0), v)

    rows, vals = slots(prob.bara)
    subi = np.atleast_1d(prob.bara.subi).astype(np.int64) - 1
    b = np.atleast_1d(prob.blc).astype(np.float64)
    if hasattr(prob, "buc") and not np.array_equal(np.atleast_1d(prob.buc).astype(np.float64), b):
        raise ValueError("only equality constraints (blc == buc) are supported")
    L = int(off[-1])
    At = sp.csc_matrix((vals, (rows, subi)), shape=(L, b.size))
    crow, cval = slots(prob.barc)
    C = np.zeros(L)
    np.add.at(C, crow, cval)
    return _finish(bardim, At, b, C)


def problem_from_sdpa(path):
    """SDPA sparse format.  With F_0, F_i and c of the file: C = -F_0, A_i = -F_i, b = -c (the convention of SDPT3's
    read_sdpa.m, which the reference's sdpa_to_txt.m uses)."""
    opener = open
    if path.endswith(".gz"):
        import gzip
        opener = gzip.open
    lines = []
    with opener(path, "rt") as f:
        for ln in f:
            if ln[:1] in '*"':
                continue
            for ch in ",{}()":
                ln = ln.replace(ch, " ")
            ln = ln.split("=")[0].split("*")[0].split('"')[0] if ("=" in ln or "*" in ln or '"' in ln) else ln
            if ln.strip():
                lines.append(ln)
    tok = " ".join(lines).split()
    m, nblk = int(float(tok[0])), int(float(tok[1]))
    sizes = np.array([int(float(t)) for t in tok[2:2 + nblk]], dtype=np.int64)
    if np.any(sizes <= 1):
        raise ValueError("SDPA diagonal / 1x1 blocks (sizes %s) are linear variables: not supported (semidefinite blocks only)"
                         % sizes[sizes <= 1].tolist())
    cvec = np.array([float(t) for t in tok[2 + nblk:2 + nblk + m]])
    ent = np.array([float(t) for t in tok[2 + nblk + m:]]).reshape(-1, 5)
    matno, blkno = ent[:, 0].astype(np.int64), ent[:, 1].astype(np.int64) - 1
    i, j = ent[:, 2].astype(np.int64) - 1, ent[:, 3].astype(np.int64) - 1
    v = -ent[:, 4]
    off = _svec_offsets(sizes)
    lo, hi = np.minimum(i, j), np.maximum(i, j)
    slot = off[blkno] + hi * (hi + 1) // 2 + lo
    val = np.where(i != j, v * SQRT2, v)
    L = int(off[-1])
    isC = matno == 0
    C = np.zeros(L)
    np.add.at(C, slot[isC], val[isC])
    At = sp.csc_matrix((val[~isC], (slot[~isC], matno[~isC] - 1)), shape=(L, m))
    return _finish(sizes, At, -cvec, C)


def write_txt(problem, directory):
    """The input directory of cuadmm_exe / Problem.from_txt (src/utils/io.cu, src/problem.cu): 0-based triplets."""
    os.makedirs(directory, exist_ok=True)
    p = problem
    with open(os.path.join(directory, "blk.txt"), "w") as f:
        for n in np.asarray(p.blk_vals).tolist():
            f.write("%d\n" % n)
    with open(os.path.join(directory, "con_num.txt"), "w") as f:
        f.write("%d\n" % p.con_num)
    cp = np.asarray(p.At_csc_col_ptrs)
    cols = np.repeat(np.arange(p.con_num), np.diff(cp))
    with open(os.path.join(directory, "At.txt"), "w") as f:
        for r, c, v in zip(np.asarray(p.At_csc_row_ids).tolist(), cols.tolist(), np.asarray(p.At_csc_vals).tolist()):
            f.write("%d %d %.17g\n" % (r, c, v))
    for name, idx, val in (("b.txt", p.b_indices, p.b_vals), ("C.txt", p.C_indices, p.C_vals)):
        with open(os.path.join(directory, name), "w") as f:
            for r, v in zip(np.asarray(idx).tolist(), np.asarray(val).tolist()):
                f.write("%d 0 %.17g\n" % (r, v))


def load_any(path, blk=None):
    """Picks the converter by file type / content: .dat-s(.gz) -> SDPA; .mat with `prob` -> MOSEK, with K -> SeDuMi,
    with At/C/b only -> svec form (needs blk)."""
    if path.endswith(".dat-s") or path.endswith(".dat-s.gz"):
        return problem_from_sdpa(path)
    keys = set(k for k in sio.whosmat(path))
    names = {k[0] for k in keys}
    if "prob" in names:
        return problem_from_mosek_mat(path)
    if "K" in names:
        return problem_from_sedumi_mat(path)
    if {"At", "C", "b"} <= names:
        if blk is None:
            raise ValueError("%s holds At/C/b in svec form: pass the block sizes (blk)" % path)
        return problem_from_svec_mat(path, blk)
    raise ValueError("%s: unrecognised contents %s" % (path, sorted(names)))


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="Convert a SeDuMi / MOSEK / svec-form .mat or an SDPA .dat-s file to the TXT directory read by cuadmm_exe")
    ap.add_argument("input")
    ap.add_argument("output_dir")
    ap.add_argument("--blk", help="blk.txt (one block size per line) for svec-form .mat files")
    a = ap.parse_args(argv)
    blk = None
    if a.blk:
        with open(a.blk) as f:
            blk = [int(t.split()[-1]) for t in f.read().splitlines() if t.strip()]
    p = load_any(a.input, blk)
    write_txt(p, a.output_dir)
    print("wrote %s: %d blocks, vec_len %d, %d constraints, nnz(At) %d" % (a.output_dir, p.mat_num, p.vec_len, p.con_num, p.At_nnz))


if __name__ == "__main__":
    main()
