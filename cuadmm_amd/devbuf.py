"""Device buffers through the C ABI (cuadmm_dev_malloc / cuadmm_memcpy_*): what the op-level entry points (cuadmm_op_*,
cuadmm_psd_plan_project) take as arguments.  Host-side utility of the ctypes mirror; no torch needed."""
import ctypes as C

import numpy as np

from ._lib import check, load


class Dev:
    """numpy array mirrored in device memory via cuadmm_dev_malloc / memcpy."""

    def __init__(self, arr=None, shape=None, dtype=np.float64):
        self.lib = load()
        if arr is not None:
            arr = np.ascontiguousarray(arr)
            shape, dtype = arr.shape, arr.dtype
        self.shape, self.dtype = tuple(np.atleast_1d(shape)) if not isinstance(shape, tuple) else shape, np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        check(self.lib.cuadmm_dev_malloc(C.byref(p), max(self.nbytes, 8)))
        self.ptr = p
        if arr is not None and self.nbytes:
            check(self.lib.cuadmm_memcpy_h2d(self.ptr, arr.ctypes.data_as(C.c_void_p), self.nbytes))

    def get(self):
        out = np.empty(self.shape, self.dtype)
        check(self.lib.cuadmm_dev_sync())
        if self.nbytes:
            check(self.lib.cuadmm_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes))
        return out

    def __del__(self):
        try:
            if self.ptr:
                self.lib.cuadmm_dev_free(self.ptr)
                self.ptr = None
        except Exception:
            pass
