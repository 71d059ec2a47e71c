"""Small helpers for the GPU parity tests: device buffers through the C ABI (no torch needed)."""
import ctypes as C

import numpy as np

import cuadmm_amd
from cuadmm_amd._lib import check


from cuadmm_amd.devbuf import Dev  # noqa: E402,F401  (the tests' device buffers are the package's)


def psd_project_gpu(x, blk, eig_rank=0):
    lib = cuadmm_amd.load()
    blk = np.ascontiguousarray(blk, dtype=np.int32)
    din = Dev(np.ascontiguousarray(x, dtype=np.float64))
    dout = Dev(shape=(x.size,), dtype=np.float64)
    check(lib.cuadmm_op_psd_project_ex(din.ptr, dout.ptr, blk.ctypes.data_as(C.c_void_p), int(blk.size), int(eig_rank), None, None))
    return dout.get()


def batch_eig_gpu(mats):
    """mats: (count, n, n) symmetric -> (W (count,n) ascending, V (count,n,n) with V[i][:,k] eigenvector k, info)."""
    lib = cuadmm_amd.load()
    count, n, _ = mats.shape
    colmajor = np.ascontiguousarray(np.swapaxes(mats, 1, 2))      # element (r,c) at c*n+r
    dm = Dev(colmajor)
    dw = Dev(shape=(count, n), dtype=np.float64)
    di = Dev(np.zeros(count, np.int32))
    check(lib.cuadmm_op_batch_eig(dm.ptr, dw.ptr, di.ptr, n, count, None))
    V = np.swapaxes(dm.get(), 1, 2)
    return dw.get(), V, di.get()


def problem_to_amd(p):
    """oracle Problem -> cuadmm_amd.Problem"""
    return cuadmm_amd.Problem(p.vec_len, p.con_num, p.blk, p.At_col_ptrs, p.At_row_ids, p.At_vals,
                              p.b_idx, p.b_vals, p.C_idx, p.C_vals)
