"""One-wavefront-per-block sign kernel (n <= 32): projection time of `count` Gaussian blocks (HIP-event free: wall over
`reps` back-to-back launches); with CUADMM_PSD_DEBUG=1 the kernel prints its phase cycles.

    CUADMM_PSD_W32_PAD=<bytes> python tools/probe_w32_occ.py [n=32] [count=10000] [reps=20]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuadmm_amd
from cuadmm_amd._lib import check
from tests.helpers import Dev

lib = cuadmm_amd.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
count = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rng = np.random.default_rng(1)
L = n * (n + 1) // 2
x = rng.standard_normal(count * L)
blk = np.full(count, n, np.int32)
din, dout = Dev(x), Dev(shape=(x.size,))
dsteps = Dev(np.zeros(count, np.int32))
bp = blk.ctypes.data_as(C.c_void_p)
for _ in range(3):
    check(lib.cuadmm_op_psd_project_steps(din.ptr, dout.ptr, bp, count, dsteps.ptr, None))
check(lib.cuadmm_dev_sync())
t0 = time.time()
for _ in range(reps):
    check(lib.cuadmm_op_psd_project_steps(din.ptr, dout.ptr, bp, count, dsteps.ptr, None))
check(lib.cuadmm_dev_sync())
dt = (time.time() - t0) / reps
print("pad %s n=%d count=%d: %.1f us per projection call (incl. plan build), steps mean %.2f"
      % (os.environ.get("CUADMM_PSD_W32_PAD", "0"), n, count, dt * 1e6, dsteps.get().mean()), flush=True)
