"""Projection-only micro-benchmark: feeds Xb ~ N(0,1)^L to the fused kernel."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import cuadmm_amd
from cuadmm_amd._lib import check
from tests.helpers import Dev
lib = cuadmm_amd.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
count = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
blk = np.full(count, n, np.int32)
L = count * n * (n + 1) // 2
x = np.random.default_rng(0).standard_normal(L)
din, dout = Dev(x), Dev(shape=(L,))
for r in range(reps):
    t = time.time()
    check(lib.cuadmm_op_psd_project(din.ptr, dout.ptr, blk.ctypes.data_as(C.c_void_p), count, None))
    check(lib.cuadmm_dev_sync())
    print("rep", r, "n", n, "count", count, "wall ms %.3f" % ((time.time() - t) * 1e3), flush=True)
