"""Which blocks differ between the two variants of the one-workgroup sign kernel (CUADMM_PSD_LDS_TRIPLE = 1 / 0), and their step counts."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cuadmm_amd
from cuadmm_amd._lib import check
from cuadmm_amd.devbuf import Dev
from oracle import cuadmm_oracle as orc

lib = cuadmm_amd.load()
rng = np.random.default_rng(5)
blk = np.array([55] * 40 + [45] * 30 + [64, 33, 48, 49], np.int32)
bidx = orc.BlockIndex(blk)
x = rng.standard_normal(int(bidx.off[-1]))
for k in range(0, blk.size, 3):
    n = int(blk[k]); U = rng.standard_normal((n, 3)); G = rng.standard_normal((n, n))
    x[int(bidx.off[k]):int(bidx.off[k + 1])] = orc.BlockIndex([n]).pack([(U @ U.T + 1e-12 * (G + G.T))[None]])


def run(triple):
    os.environ["CUADMM_PSD_LDS_TRIPLE"] = triple
    din = Dev(x); dout = Dev(shape=(x.size,), dtype=np.float64); dst = Dev(np.zeros(blk.size, np.int32))
    check(lib.cuadmm_op_psd_project_ex(din.ptr, dout.ptr, blk.ctypes.data_as(C.c_void_p), int(blk.size), 0, dst.ptr, None))
    return dout.get(), dst.get()


a, sa = run("1")
b, sb = run("0")
ref = orc.psd_project_svec(bidx, x)
for k in range(blk.size):
    sl = slice(int(bidx.off[k]), int(bidx.off[k + 1]))
    d = np.max(np.abs(a[sl] - b[sl]))
    if d > 0 or sa[k] != sb[k]:
        print("block %3d n %2d moment-like %d: steps %d / %d, max diff %.3e, err vs oracle %.2e / %.2e" % (k, blk[k], k % 3 == 0, sa[k], sb[k], d,
              np.max(np.abs(a[sl] - ref[sl])), np.max(np.abs(b[sl] - ref[sl]))))
print("steps triple", sa[:12], "two-matrix", sb[:12])
print("max err vs oracle", np.max(np.abs(a - ref)), np.max(np.abs(b - ref)))
