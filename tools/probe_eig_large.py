"""cuadmm_op_batch_eig (explicit eigendecomposition, cusolver.h:76-95 contract) at large n: time and accuracy vs LAPACK."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import batch_eig_gpu
for n in [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2000]:
    rng = np.random.default_rng(n)
    G = rng.standard_normal((1, n, n)); A = (G + np.swapaxes(G, 1, 2)) / 2
    t = time.time(); W, V, info = batch_eig_gpu(A); dt = time.time() - t
    w = np.linalg.eigvalsh(A[0]); nrm = np.abs(w).max()
    print("n=%d: %.2f s  info %d  eig err %.2e  resid %.2e  orth %.2e" % (
        n, dt, int(info[0]), np.abs(W[0] - w).max() / nrm,
        np.abs(V[0] * W[0][None, :] @ V[0].T - A[0]).max() / nrm, np.abs(V[0].T @ V[0] - np.eye(n)).max()), flush=True)
