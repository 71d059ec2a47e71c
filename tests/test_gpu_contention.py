"""The inter-workgroup protocols under a SECOND PROCESS that saturates the GPU (What's weak #10).

Two kernels rest on workgroups being co-resident: the four-workgroups-per-row tail kernel (csrc/tail_solve.hip,
ts_onepass_group_kernel: members of a row exchange their parts through sentinel slots) and the one-launch matrix-sign kernel
(csrc/psd_large.hip, lg_sign_cluster_kernel: the workgroups of a member meet at a barrier).  HIP promises neither dispatch order nor
residency; both kernels carry a give-up path (a raised counter and CUADMM_ERR_FACTOR / CUADMM_ERR_EIG instead of a hang).  The
single-process tests never make the dispatcher share the chip.  Here a child process loops C2-like solves (every CU busy with
persistent 16-wavefront workgroups) while this process runs the K > 18 432 tail solves and one-launch solves: results must equal
the quiet run's, bit for bit, and no give-up counter may move (an error return would fail the calls)."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest
import scipy.linalg as sl
import ctypes as C

import cuadmm_amd
from cuadmm_amd._lib import check
from tests.conftest import load_npz_problem
from tests.helpers import problem_to_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HOG = r'''
import sys, time
sys.path.insert(0, %r)
import cuadmm_amd
from cuadmm_amd import synthetic
p = synthetic.make_synthetic([32] * 6000, dense_C=True)
prob = cuadmm_amd.Problem(p.vec_len, p.con_num, p.blk, p.At_col_ptrs, p.At_row_ids, p.At_vals, p.b_idx, p.b_vals, p.C_idx, p.C_vals)
s = cuadmm_amd.SDPSolver(verbose=False)
s.init_problem(prob)
print("ready", flush=True)
t0 = time.time()
while time.time() - t0 < float(sys.argv[1]):
    s.solve(400, 0.0, 0, 50, 100, 0, 1.05, if_first=False)
'''


def _cluster_solve(prob):
    s = cuadmm_amd.SDPSolver(verbose=False, options={"psd_lg_cluster": 1})
    s.init_problem(prob)
    s.solve(12, 0.0, 0, 50, 100, 11000, 1.05)
    return np.array(s.info_arr("pobj")), s.X


def _tail(L, D, z):
    got = np.stack([z, z]).copy()
    k = z.size
    check(cuadmm_amd.load().cuadmm_op_tail_solve(L.ctypes.data_as(C.c_void_p), D.ctypes.data_as(C.c_void_p), k, got.ctypes.data_as(C.c_void_p), 2))
    return got


def test_workgroup_protocols_with_a_second_process_on_the_gpu():
    prob = problem_to_amd(load_npz_problem("PlanarHand_N=1_MOMENT"))
    k = 18500
    rng = np.random.default_rng(9)
    Lm = rng.random((k, k), dtype=np.float32).astype(np.float64)
    Lm -= 0.5
    Lm *= 1.0 / np.sqrt(k)
    Lm = np.tril(Lm, -1)
    Lm[np.diag_indices(k)] = 1.0
    D = rng.uniform(0.1, 2.0, k)
    z = rng.standard_normal(k)
    quiet_pobj, quiet_X = _cluster_solve(prob)
    quiet_tail = _tail(Lm, D, z)
    ref = sl.solve_triangular(Lm.T, sl.solve_triangular(Lm, z, lower=True, unit_diagonal=True) / D, lower=False, unit_diagonal=True)
    assert np.linalg.norm(quiet_tail[0] - ref) <= 1e-13 * np.linalg.norm(ref)
    hog = subprocess.Popen([sys.executable, "-c", HOG % ROOT, "60"], stdout=subprocess.PIPE, text=True,
                           env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    try:
        assert hog.stdout.readline().strip() == "ready"
        time.sleep(0.5)
        t0, rounds = time.time(), 0
        while time.time() - t0 < 10.0 or rounds < 3:
            assert hog.poll() is None, "the competing process ended early"
            pobj, X = _cluster_solve(prob)                 # raises on a give-up (CUADMM_ERR_EIG)
            assert np.array_equal(pobj, quiet_pobj) and np.array_equal(X, quiet_X)
            got = _tail(Lm, D, z)                          # raises on a lost exchange (CUADMM_ERR_FACTOR)
            assert np.array_equal(got, quiet_tail)
            rounds += 1
        assert rounds >= 3
    finally:
        hog.kill()
        hog.wait()
