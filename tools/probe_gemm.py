import sys, os, time
import numpy as np, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuadmm_amd
from cuadmm_amd._lib import check
from tests.helpers import Dev
lib = cuadmm_amd.load()
for n in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n)); A = (A + A.T) / 2
    B = rng.standard_normal((n, n)); B = (B + B.T) / 2
    E = rng.standard_normal((n, n))
    dA, dB, dE, dC = Dev(A), Dev(B), Dev(E), Dev(shape=(n, n))
    check(lib.cuadmm_op_gemm_sym(n, dA.ptr, dB.ptr, 0.75, -1.25, dE.ptr, dC.ptr, None)); check(lib.cuadmm_dev_sync())
    got = dC.get(); ref = 0.75 * A @ B - 1.25 * E
    t = time.time()
    for _ in range(30): check(lib.cuadmm_op_gemm_sym(n, dA.ptr, dB.ptr, 1.0, 0.0, None, dC.ptr, None))
    check(lib.cuadmm_dev_sync()); dt = (time.time() - t) / 30
    print("n %d  max err %.2e  nan %d  %.3f ms  %.1f TF" % (n, np.nanmax(np.abs(got - ref)), np.isnan(got).sum(), dt * 1e3, 2 * n**3 / dt * 1e-12))
    if np.isnan(got).sum() or np.nanmax(np.abs(got - ref)) > 1e-9:
        bad = np.argwhere(~(np.abs(got - ref) < 1e-9)); print("bad entries", bad.shape[0], bad[:10].tolist())
