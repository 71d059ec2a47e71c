import sys, os, ctypes as C, numpy as np, pickle
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import sign_schedule_sim as T
from tests.conftest import load_npz_problem
name = sys.argv[1]; iters = int(sys.argv[2])
cache = "/tmp/spec_%s_%d.pkl" % (name, iters)
if os.path.exists(cache): samp = pickle.load(open(cache, "rb"))
else:
    p = load_npz_problem(name)
    samp = T.spectra_from_oracle(p, iters, max(1, iters // 5))
    pickle.dump(samp, open(cache, "wb"))
lib = C.CDLL("/tmp/cm/libsim.so")
def run(sp, mode):
    v = np.ascontiguousarray(np.abs(sp), dtype=np.float64).copy(); err = C.c_double()
    return lib.sim(v.ctypes.data_as(C.c_void_p), int(v.size), mode, C.byref(err)), err.value
for k, eigs, n1 in samp:
    rows = []
    for w in eigs:
        for row in w:
            nf = np.sqrt(np.sum(row * row))
            if nf == 0 or len(row) <= 64: continue
            sp = row / nf
            rows.append((len(row), run(sp, 9)[0], run(sp, 1)[0], run(sp, 17)[0], max(run(sp, 1)[1], run(sp, 17)[1])))
    r = np.array(rows, float)
    print("it %5d: %4d blocks n>64: lagged steps none %.2f (max %d)  capped %.2f (max %d)  clean %.2f (max %d)  err %.1e" % (k, len(r), r[:,1].mean(), r[:,1].max(), r[:,2].mean(), r[:,2].max(), r[:,3].mean(), r[:,3].max(), r[:,4].max()))
