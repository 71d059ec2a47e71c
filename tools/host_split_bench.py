"""Host leading sweeps of a split factor: time per forward / backward sweep (CUADMM_HOST_THREADS=1 for the serial figure).
    python tools/host_split_bench.py <fixture>"""
import sys, os, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuadmm_amd
from cuadmm_amd._lib import check
from tests.conftest import load_npz_problem
import scipy.sparse as sp
lib = cuadmm_amd.load()
P = lambda a: a.ctypes.data_as(C.c_void_p)
p = load_npz_problem(sys.argv[1])
At = sp.csc_matrix((p.At_vals, p.At_row_ids, p.At_col_ptrs), shape=(p.vec_len, p.con_num))
nrm = np.maximum(1.0, np.sqrt(np.asarray(At.multiply(At).sum(axis=0)).ravel()))
A = (At @ sp.diags(1.0 / nrm)).T.tocsc(); A.sort_indices()
cp, ri, vx = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data.astype(np.float64)
h = C.c_void_p()
t = time.time(); check(lib.cuadmm_aat_create_split(p.con_num, p.vec_len, P(cp), P(ri), P(vx), 1e-15, 32768, C.byref(h))); tf = time.time() - t
m = p.con_num; k = lib.cuadmm_aat_tail_k(h)
print("m %d tail_k %d nnz(L) %d factor %.2f s, pool threads %d" % (m, k, lib.cuadmm_aat_factor_nnz(h), tf, lib.cuadmm_host_pool_threads()))
rhs = np.random.default_rng(0).standard_normal(m)
for rep in range(3):
    x = rhs.copy()
    t0 = time.perf_counter()
    for _ in range(20): check(lib.cuadmm_aat_solve_leading_forward(h, k, P(x)))
    t1 = time.perf_counter()
    for _ in range(20): check(lib.cuadmm_aat_solve_leading_backward(h, k, P(x)))
    t2 = time.perf_counter()
    print("forward %.3f ms  backward %.3f ms" % ((t1 - t0) / 20 * 1e3, (t2 - t1) / 20 * 1e3))
