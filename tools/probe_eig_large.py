"""cuadmm_op_batch_eig (explicit eigendecomposition, cusolver.h:76-95 contract) at large n: time and accuracy vs LAPACK.
    python tools/probe_eig_large.py [n ...]"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import batch_eig_gpu


def matrix(n, kind, rng):
    if kind == "randn":
        G = rng.standard_normal((n, n)); return (G + G.T) / 2
    if kind == "rank1":
        v = rng.standard_normal(n); return np.outer(v, v)
    if kind == "lowrank":
        U = rng.standard_normal((n, 5)); G = rng.standard_normal((n, n)); return U @ U.T + 1e-7 * (G + G.T)
    if kind == "identity":
        return np.eye(n)
    if kind == "zero":
        return np.zeros((n, n))
    if kind == "diag":
        return np.diag(rng.standard_normal(n))
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    if kind == "graded":
        w = np.logspace(-14, 0, n) * np.where(np.arange(n) % 2, 1.0, -1.0)
    elif kind == "clustered":      # five eigenvalues, each n / 5 times
        w = np.repeat(np.array([-2.0, -1e-3, 0.0, 1.0, 1.0 + 1e-9]), (n + 4) // 5)[:n]
    elif kind == "tight":          # pairs 1e-13 apart
        w = np.repeat(np.linspace(-1, 1, (n + 1) // 2), 2)[:n] + np.tile([0.0, 1e-13], (n + 1) // 2)[:n]
    return (Q * w) @ Q.T


kinds = ["randn", "rank1", "lowrank", "identity", "zero", "diag", "graded", "clustered", "tight"]
for n in [int(a) for a in sys.argv[1:]] or [130, 257, 500, 1024, 2000]:
    for kind in kinds if n <= 1100 else ["randn", "lowrank", "clustered"]:
        rng = np.random.default_rng(n)
        A = matrix(n, kind, rng); A = (A + A.T) / 2
        t = time.time(); W, V, info = batch_eig_gpu(A[None]); dt = time.time() - t
        w = np.linalg.eigvalsh(A); nrm = max(np.abs(w).max(), 1e-300)
        print("n=%4d %-9s: %7.3f s  info %d  eig err %.2e  resid %.2e  orth %.2e  ascending %s" % (
            n, kind, dt, int(info[0]), np.abs(W[0] - w).max() / nrm,
            np.abs((V[0] * W[0][None, :]) @ V[0].T - A).max() / nrm, np.abs(V[0].T @ V[0] - np.eye(n)).max(), bool(np.all(np.diff(W[0]) >= 0))), flush=True)
