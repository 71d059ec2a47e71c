"""GPU tail solve (inverse of a dense unit-lower triangle by recursive doubling + two GEMVs) against scipy."""
import sys, os, time, ctypes as C
import numpy as np, scipy.linalg as sl
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuadmm_amd
from cuadmm_amd._lib import check
lib = cuadmm_amd.load()
P = lambda a: a.ctypes.data_as(C.c_void_p)
for k in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(k)
    L = np.tril(rng.standard_normal((k, k)) * (0.5 / np.sqrt(k)), -1) + np.eye(k)
    D = rng.uniform(0.1, 2.0, k)
    z = rng.standard_normal((3, k))
    ref = np.stack([sl.solve_triangular(L.T, sl.solve_triangular(L, zi, lower=True, unit_diagonal=True) / D, lower=False, unit_diagonal=True) for zi in z])
    got = z.copy()
    t = time.time(); check(lib.cuadmm_op_tail_solve(P(L), P(D), k, P(got), 3)); dt = time.time() - t
    print("k %6d  rel err %.2e  (build + 3 solves %.2f s)" % (k, np.linalg.norm(got - ref) / np.linalg.norm(ref), dt), flush=True)
import scipy.sparse as sp
for k in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(k + 1)
    G = rng.standard_normal((k, k)) / np.sqrt(k)
    S = G @ G.T + 0.5 * np.eye(k)
    Sl = sp.csr_matrix(np.tril(S))
    rp, ci, vv = Sl.indptr.astype(np.int64), Sl.indices.astype(np.int32), Sl.data.astype(np.float64)
    z = rng.standard_normal((2, k)); ref = np.linalg.solve(S, z.T).T
    got = z.copy()
    t = time.time(); check(lib.cuadmm_op_tail_factor_solve(P(rp), P(ci), P(vv), k, P(got), 2)); dt = time.time() - t
    print("factor+solve k %6d  rel err %.2e  (%.2f s)" % (k, np.linalg.norm(got - ref) / np.linalg.norm(ref), dt), flush=True)
