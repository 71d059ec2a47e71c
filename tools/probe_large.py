"""Accuracy / speed of the large-block projection (matrix-sign path) against numpy eigh."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import psd_project_gpu
from oracle import cuadmm_oracle as orc

for spec in sys.argv[1:]:
    n = int(spec.split(":")[0])
    kind = spec.split(":")[1] if ":" in spec else "randn"
    rng = np.random.default_rng(n)
    if kind == "randn":
        M = rng.standard_normal((n, n)); M = (M + M.T) / 2
    elif kind == "lowrank":      # rank-5 PSD minus small indefinite noise (typical late-ADMM iterate)
        U = rng.standard_normal((n, 5)); M = U @ U.T + 1e-6 * (lambda G: (G + G.T) / 2)(rng.standard_normal((n, n)))
    elif kind == "graded":
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        lam = np.concatenate([10.0 ** rng.uniform(-14, 0, n // 2), -10.0 ** rng.uniform(-14, 0, n - n // 2)])
        M = (Q * lam) @ Q.T; M = (M + M.T) / 2
    blk = np.array([n], np.int32)
    bi = orc.BlockIndex(blk)
    x = bi.pack([M[None]])
    t = time.time(); got = psd_project_gpu(x, blk); t1 = time.time() - t
    t = time.time(); got = psd_project_gpu(x, blk); t2 = time.time() - t
    t = time.time(); ref = orc.psd_project_svec(bi, x); tc = time.time() - t
    print("n %5d %-8s wall %.1f ms (2nd call) | cpu eigh %.1f ms | max abs err %.2e (||X||_2 ~ %.2e)" %
          (n, kind, t2 * 1e3, tc * 1e3, np.max(np.abs(got - ref)), np.linalg.norm(M, 2)), flush=True)
