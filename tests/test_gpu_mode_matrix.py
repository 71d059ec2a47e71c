"""The engine's mode matrix on ONE small coupled problem: every behavioural switch is an option of the
C ABI (cuadmm_set_option; include/cuadmm_amd.h), and every combination the planner can be steered into must reproduce the
default engine's trajectory (<= 1e-9 relative per iteration, sigma exact, X / S <= 1e-9) -- sGS phase, the switch and the
ADMM phase.  Covered: fusion on / off, local rows in / out of the projection kernels, y-solve on the host / device, host-only
factor / forced GPU tail (with and without the device-side leading sweeps), mapped result buffer on / off, every kernel choice
of the 9 <= n <= 64 classes, longest-block-first off, the sGS second half in one or two passes -- and the same with two
shards (general all-reduce path) for the combinations that change what is replicated."""
import itertools
import threading

import numpy as np
import pytest

import cuadmm_amd
from tests.test_gpu_fused import _coupled_problem
from tests.test_gpu_sharded import HostAllReduce

pytestmark = pytest.mark.gpu

NAMES = ("errRp", "errRd", "pobj", "dobj", "relgap")
BLK = [32] * 30 + [45] * 6 + [12] * 20 + [6] * 12 + [20] * 10 + [70] * 2
SOLVE = (24, 0.0, 0, 50, 7, 9, 1.05)          # 8 sGS iterations, the switch at 9, ADMM after it; sigma moves every 7


@pytest.fixture(scope="module")
def reference():
    prob = _coupled_problem(BLK, n_couple=30)
    s = cuadmm_amd.SDPSolver(verbose=False)
    s.init_problem(prob)
    s.solve(*SOLVE)
    return prob, {nm: s.info_arr(nm) for nm in NAMES + ("sig",)}, s.X, s.S


def _check(s, ref):
    _, info, X, S = ref
    for nm in NAMES:
        got = s.info_arr(nm)
        assert np.max(np.abs(got - info[nm]) / (1e-12 / 1e-9 + np.abs(info[nm]))) <= 1e-9, nm
    assert np.array_equal(s.info_arr("sig"), info["sig"])
    assert np.max(np.abs(s.X - X)) <= 1e-9 * (1 + np.max(np.abs(X)))
    assert np.max(np.abs(s.S - S)) <= 1e-9 * (1 + np.max(np.abs(S)))


GRID = [dict(zip(("fuse", "fuse_rows", "host_solve", "tail_k", "mapped_out"), v))
        for v in itertools.product((0, 1), (0, 1), (0, 1), (-1, 0, 64), (0, 1)) if not (v[0] == 0 and v[1] == 0)]


@pytest.mark.parametrize("opts", GRID, ids=lambda o: "-".join("%s%d" % (k[:4], v) for k, v in o.items()))
def test_iteration_mode_matrix(opts, reference):
    s = cuadmm_amd.SDPSolver(verbose=False, options=opts)
    s.init_problem(reference[0])
    s.solve(*SOLVE)
    c = s.counters()
    if opts["tail_k"] == 64:
        assert c["tail_k"] == 64
    if opts["tail_k"] == 0:
        assert c["tail_k"] == 0
    _check(s, reference)


@pytest.mark.parametrize("opts", [{"psd_n16": 0}, {"psd_n32": 0}, {"psd_mid": 1}, {"psd_mid": 2}, {"psd_wave4_min": 1}, {"psd_w32_occ": 3},
                                  {"psd_overlap": 0}, {"lpt": 0}, {"aty_post2": 0}, {"psd_hint": 0}, {"psd_hint": 2}, {"tiny_sign": 2},
                                  {"psd_sign_min": 100}, {"psd_lg_tile": 64}, {"psd_lg_decide": 2}, {"psd_sign_sync": 0}, {"lazy_unscale": 0}],
                         ids=lambda o: "-".join("%s%d" % kv for kv in o.items()))
def test_kernel_choice_matrix(opts, reference):
    s = cuadmm_amd.SDPSolver(verbose=False, options=opts)
    s.init_problem(reference[0])
    s.solve(*SOLVE)
    _check(s, reference)


@pytest.mark.parametrize("opts", [{}, {"fuse": 0}, {"tail_k": 64}, {"tail_k": 64, "host_solve": 1}, {"host_solve": 1, "tail_k": 0}, {"mapped_out": 0, "fuse_rows": 0}],
                         ids=lambda o: "-".join("%s%d" % kv for kv in o.items()) or "default")
def test_two_shards_over_the_general_path(opts, reference):
    """The same problem on two ranks (two engines in two host threads, host-staged all-reduce): coupled constraints, so [A X | sums |
    A (S - C)] is all-reduced and the solve -- host, GPU tail, device-side sweeps -- is replicated."""
    prob = reference[0]
    world = 2
    ar = HostAllReduce(world)
    solvers = [cuadmm_amd.SDPSolver(verbose=False, rank=r, world=world, options=opts) for r in range(world)]
    errs = []

    def run(r):
        try:
            solvers[r].set_allreduce(ar.hook(r))
            solvers[r].init_problem(prob)
            solvers[r].solve(*SOLVE)
        except Exception as e:          # pragma: no cover
            errs.append(e)
            ar.barrier.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    _, info, X, S = reference
    for nm in NAMES:
        got = solvers[0].info_arr(nm)
        assert np.array_equal(got, solvers[1].info_arr(nm)), nm
        assert np.max(np.abs(got - info[nm]) / (1e-12 / 1e-9 + np.abs(info[nm]))) <= 1e-9, nm
    Xs = np.concatenate([s.X for s in solvers])
    assert np.max(np.abs(Xs - X)) <= 1e-9 * (1 + np.max(np.abs(X)))
