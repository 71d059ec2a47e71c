"""Debug helper: drive the oracle on PlanarHand and run every Xb through the GPU projection; dump inputs of
blocks where the GPU kernel reports a QL cap hit or disagrees with LAPACK."""
import sys, os, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cuadmm_amd
from oracle import cuadmm_oracle as orc
from tests.conftest import load_npz_problem
from tests.helpers import Dev
lib = cuadmm_amd.load()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 120
p = load_npz_problem("PlanarHand_N=1_MOMENT")
blk = np.ascontiguousarray(p.blk, np.int32)
bidx = orc.BlockIndex(blk)
bad = []
calls = [0]
def eig_fn(bi, xb):
    calls[0] += 1
    xb = np.ascontiguousarray(xb)
    din, dout = Dev(xb), Dev(shape=xb.shape)
    rc = lib.cuadmm_op_psd_project(din.ptr, dout.ptr, blk.ctypes.data_as(C.c_void_p), int(blk.size), None)
    got = dout.get()
    ref = orc.psd_project_svec(bi, xb)
    diff = np.abs(got - ref)
    if rc != 0 or diff.max() > 1e-10:
        for k in range(blk.size):
            sl = slice(int(bi.off[k]), int(bi.off[k + 1]))
            d = diff[sl].max()
            if d > 1e-10 or not np.all(np.isfinite(got[sl])):
                print("call", calls[0], "rc", rc, "block", k, "n", blk[k], "maxdiff", d, flush=True)
                bad.append((calls[0], k, int(blk[k]), xb[sl].copy(), got[sl].copy()))
        if rc != 0 and not bad:
            print("call", calls[0], "rc", rc, "but all blocks agree", flush=True)
    return ref
t = time.time()
s = orc.OracleSolver(eig_fn=eig_fn).init_problem(p)
print("init", time.time() - t, flush=True)
s.solve(iters, 1e-3, 0, 50, 100, 0, 1.05, verbose=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "ph_bad_blocks.npz"),
                    calls=np.array([b[0] for b in bad]), ks=np.array([b[1] for b in bad]), ns=np.array([b[2] for b in bad]),
                    **{"x%d" % i: b[3] for i, b in enumerate(bad)}, **{"g%d" % i: b[4] for i, b in enumerate(bad)})
print("bad blocks:", len(bad), "time", time.time() - t)
