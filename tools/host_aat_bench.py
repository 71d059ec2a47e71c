"""Host-side AA^T factor + permuted solve timing on a real-data problem (no GPU needed)."""
import sys, os, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuadmm_amd
from cuadmm_amd._lib import check
from tests.conftest import load_npz_problem
lib = cuadmm_amd.load()
P = lambda a: a.ctypes.data_as(C.c_void_p)
p = load_npz_problem(sys.argv[1])
import scipy.sparse as sp
At = sp.csc_matrix((p.At_vals, p.At_row_ids, p.At_col_ptrs), shape=(p.vec_len, p.con_num))
nrm = np.maximum(1.0, np.sqrt(np.asarray(At.multiply(At).sum(axis=0)).ravel()))      # get_normA
A = (At @ sp.diags(1.0 / nrm)).T.tocsc(); A.sort_indices()
cp, ri, vx = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
h = C.c_void_p()
t = time.time(); check(lib.cuadmm_aat_create(p.con_num, p.vec_len, P(cp), P(ri), P(vx), 1e-15, C.byref(h))); tf = time.time() - t
m = p.con_num
print("m %d  nnz(At) %d  nnz(L) %d  factor %.2f s" % (m, vx.size, lib.cuadmm_aat_factor_nnz(h), tf))
rhs = np.random.default_rng(0).standard_normal(m); out = np.empty(m)
for rep in range(3):
    t = time.time()
    for _ in range(20): check(lib.cuadmm_aat_solve_permuted(h, P(rhs), P(out)))
    print("solve %.3f ms" % ((time.time() - t) / 20 * 1e3))
lib.cuadmm_aat_free(h)
