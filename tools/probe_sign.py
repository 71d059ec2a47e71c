"""Adaptive matrix-sign projection on the GPU: accuracy against LAPACK and Newton-Schulz step counts per spectrum family.

    python tools/probe_sign.py [n=32] [count=10000]
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
SQ2 = float.fromhex("0x1.6a09e667f3bccp+0")


def pack(M):
    n = M.shape[-1]
    ii, jj = np.tril_indices(n)
    sc = np.where(ii == jj, 1.0, SQ2)
    return (M[:, jj, ii] * sc[None, :]).reshape(-1)


def unpack(x, n, count):
    ii, jj = np.tril_indices(n)
    sc = np.where(ii == jj, 1.0, 1.0 / SQ2)
    v = x.reshape(count, -1) * sc[None, :]
    M = np.zeros((count, n, n))
    M[:, jj, ii] = v
    M[:, ii, jj] = v
    return M


def family(name, n, count, rng):
    Q = np.linalg.qr(rng.standard_normal((count, n, n)))[0]
    if name == "gaussian":
        G = rng.standard_normal((count, n, n))
        return (G + np.swapaxes(G, 1, 2)) / 2
    if name == "graded":
        w = np.logspace(0, -10, n)[None, :] * rng.choice([-1.0, 1.0], (count, n))
    elif name == "rank_deficient":
        w = rng.standard_normal((count, n))
        w[:, : (2 * n) // 3] = 0.0
    elif name == "pm1e-13":
        w = rng.standard_normal((count, n))
        w[:, 0] = 1e-13
        w[:, 1] = -1e-13
    elif name == "moment_like":
        w = 10.0 ** rng.uniform(-16, -11, (count, n)) * rng.choice([-1.0, 1.0], (count, n))
        w[:, -2:] = rng.uniform(0.2, 1.0, (count, 2))
    else:
        raise ValueError(name)
    return (Q * w[:, None, :]) @ np.swapaxes(Q, 1, 2)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    rng = np.random.default_rng(1)
    blk = np.full(count, n, np.int32)
    for name in ["gaussian", "graded", "rank_deficient", "pm1e-13", "moment_like"]:
        M = family(name, n, count, rng)
        M = (M + np.swapaxes(M, 1, 2)) / 2
        x = pack(M)
        din, dout = Dev(x), Dev(shape=(x.size,))
        dsteps = Dev(np.zeros(count, np.int32))
        t0 = time.time()
        check(lib.cuadmm_op_psd_project_steps(din.ptr, dout.ptr, blk.ctypes.data_as(C.c_void_p), count, dsteps.ptr, None))
        check(lib.cuadmm_dev_sync())
        wall = time.time() - t0
        P = unpack(dout.get(), n, count)
        steps = dsteps.get()
        w, V = np.linalg.eigh(M)
        ref = (V * np.maximum(w, 0)[:, None, :]) @ np.swapaxes(V, 1, 2)
        nrm = np.abs(M).sum(axis=1).max(axis=1)
        err = np.abs(P - ref).max(axis=(1, 2)) / np.where(nrm > 0, nrm, 1)
        # run-to-run reproducibility
        check(lib.cuadmm_op_psd_project_steps(din.ptr, dout.ptr, blk.ctypes.data_as(C.c_void_p), count, dsteps.ptr, None))
        same = np.array_equal(unpack(dout.get(), n, count), P)
        print("n=%d %-15s blocks %d: steps mean %.2f p90 %d max %d | max err/||X||_1 %.2e | bit-identical rerun %s | wall %.1f ms"
              % (n, name, count, steps.mean(), np.percentile(steps, 90), steps.max(), err.max(), same, wall * 1e3), flush=True)


if __name__ == "__main__":
    main()
