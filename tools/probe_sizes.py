"""Times the projection for a list of block sizes (count blocks each) -- wall time of repeated launches."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import cuadmm_amd
from cuadmm_amd._lib import check
from tests.helpers import Dev
from oracle import cuadmm_oracle as orc
lib = cuadmm_amd.load()
for spec in sys.argv[1:]:
    n, count = [int(x) for x in spec.split("x")]
    blk = np.full(count, n, np.int32)
    L = count * n * (n + 1) // 2
    x = np.random.default_rng(0).standard_normal(L)
    din, dout = Dev(x), Dev(shape=(L,))
    ts = []
    for r in range(4):
        t = time.time()
        check(lib.cuadmm_op_psd_project(din.ptr, dout.ptr, blk.ctypes.data_as(C.c_void_p), count, None))
        check(lib.cuadmm_dev_sync())
        ts.append((time.time() - t) * 1e3)
    got = dout.get()
    k = min(count, 8)
    sub = slice(0, k * n * (n + 1) // 2)
    ref = orc.psd_project_svec(orc.BlockIndex(blk[:k]), x[sub])
    print("n %4d count %6d  wall ms min %.3f  (%.2f us/block)  maxdiff %.2e" % (n, count, min(ts), min(ts) * 1e3 / count, np.max(np.abs(got[sub] - ref))), flush=True)
