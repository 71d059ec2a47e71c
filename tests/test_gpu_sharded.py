"""Block-sharded engine (world=2) on ONE GPU: two engines in two host threads, the all-reduce hook
implemented through a host staging buffer.  Checks the sharding arithmetic of SURVEY.md section 8e
(contiguous block ranges, partial A*v summed before the replicated host solve) against the
single-engine run.  The RCCL transport itself is exercised by bench.py --gpus N on multi-GPU nodes."""
import ctypes as C
import threading

import numpy as np
import pytest

import cuadmm_amd
from cuadmm_amd._lib import check
from cuadmm_amd.synthetic import make_synthetic

pytestmark = pytest.mark.gpu


def _amd(p):
    return cuadmm_amd.Problem(p.vec_len, p.con_num, p.blk, p.At_col_ptrs, p.At_row_ids, p.At_vals,
                              p.b_idx, p.b_vals, p.C_idx, p.C_vals)


class HostAllReduce:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.bufs = [None] * world
        self.lib = cuadmm_amd.load()

    def hook(self, rank):
        def fn(ptr, count, stream):
            check(self.lib.cuadmm_dev_sync())
            mine = np.empty(count)
            check(self.lib.cuadmm_memcpy_d2h(mine.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), count * 8))
            self.bufs[rank] = mine
            self.barrier.wait()
            total = np.zeros(count)
            for r in range(self.world):            # fixed order: every rank gets bit-identical sums
                total += self.bufs[r]
            self.barrier.wait()
            check(self.lib.cuadmm_memcpy_h2d(C.c_void_p(ptr), total.ctypes.data_as(C.c_void_p), count * 8))
        return fn


@pytest.mark.parametrize("mode", ["owned_constraints", "replicated_solve"])
@pytest.mark.parametrize("sw,iters", [(0, 25), (10 ** 9, 12), (8, 20)])
def test_two_shards_match_single_engine(sw, iters, mode, monkeypatch):
    """Every constraint of the synthetic problem touches one block, hence one rank: by default each rank then keeps only
    its own constraints and the ranks exchange four scalars per iteration ("owned constraints"); with
    option local_constraints = 0 the general path runs (A*X all-reduced, solve replicated).  Both must agree with the
    single engine."""
    opts = {"local_constraints": 0} if mode == "replicated_solve" else None
    blk = [32] * 40 + [7] * 30 + [15] * 21 + [40, 3, 3, 28]
    rng = np.random.default_rng(2)
    blk = list(np.array(blk)[rng.permutation(len(blk))])
    p = make_synthetic(blk, cons_per_block=3, seed=11)
    ref = cuadmm_amd.SDPSolver(verbose=False)
    ref.init_problem(_amd(p))
    ref.solve(iters, 0.0, 0, 50, 100, sw, 1.05)

    world = 2
    ar = HostAllReduce(world)
    solvers = [cuadmm_amd.SDPSolver(verbose=False, rank=r, world=world, options=opts) for r in range(world)]
    errs = []

    def run(r):
        try:
            solvers[r].set_allreduce(ar.hook(r))
            solvers[r].init_problem(_amd(p))
            solvers[r].solve(iters, 0.0, 0, 50, 100, sw, 1.05)
        except Exception as e:          # pragma: no cover
            errs.append(e)
            ar.barrier.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    shards = [s.shard() for s in solvers]
    assert shards[0][0] == 0 and shards[0][1] == shards[1][0] and shards[1][1] == p.vec_len
    assert shards[0][3] == shards[1][2] and 0 < shards[0][3] < len(blk)
    for nm in ("errRp", "errRd", "pobj", "dobj", "relgap", "sig"):
        a, b0, b1 = ref.info_arr(nm), solvers[0].info_arr(nm), solvers[1].info_arr(nm)
        assert np.array_equal(b0, b1), nm                                   # replicated host state is identical
        assert np.max(np.abs(a - b0) / (1e-12 + np.abs(a))) <= 1e-9, nm     # only the summation order differs
    X = np.concatenate([s.X for s in solvers])
    S = np.concatenate([s.S for s in solvers])
    assert np.max(np.abs(X - ref.X)) <= 1e-9 * (1 + np.max(np.abs(ref.X)))
    assert np.max(np.abs(S - ref.S)) <= 1e-9 * (1 + np.max(np.abs(ref.S)))
    assert np.max(np.abs(solvers[0].y - ref.y)) <= 1e-8 * (1 + np.max(np.abs(ref.y)))


def test_ranks_agree_on_batching_when_only_one_shard_can_batch():
    """Heterogeneous shards under owned-constraints sharding: rank 0 holds only 32 x 32 blocks (closed, one tile geometry: it
    could run several iterations per launch), rank 1 a mix of geometries (it cannot).  A rank that batches all-reduces 4 K
    scalars once per batch, one that does not 4 per iteration: the decision is taken together at the start of the solve
    (Solver::batch_agree), here against it on BOTH ranks -- and the trajectory is the single engine's."""
    blk = [32] * 600 + [45] * 100 + [12] * 400 + [28] * 100      # >= 256 blocks per rank: the y-solve runs on the device, blocks are closed
    p = make_synthetic(blk, cons_per_block=3, seed=5)
    iters = 40
    ref = cuadmm_amd.SDPSolver(verbose=False)
    ref.init_problem(_amd(p))
    ref.solve(iters, 0.0, 0, 50, 100, 0, 1.05)
    world = 2
    ar = HostAllReduce(world)
    solvers = [cuadmm_amd.SDPSolver(verbose=False, rank=r, world=world) for r in range(world)]
    errs = []

    def run(r):
        try:
            solvers[r].set_allreduce(ar.hook(r))
            solvers[r].init_problem(_amd(p))
            solvers[r].solve(iters, 0.0, 0, 50, 100, 0, 1.05)
        except Exception as e:          # pragma: no cover
            errs.append(e)
            ar.barrier.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    shards = [s.shard() for s in solvers]
    assert all(b == 32 for b in blk[:shards[0][3]])                          # rank 0: 32 x 32 blocks only
    assert [s.counters()["batch_launches"] for s in solvers] == [0.0, 0.0]   # neither rank batched
    for nm in ("errRp", "errRd", "pobj", "dobj", "relgap", "sig"):
        a, b0, b1 = ref.info_arr(nm), solvers[0].info_arr(nm), solvers[1].info_arr(nm)
        assert np.array_equal(b0, b1), nm
        assert np.max(np.abs(a - b0) / (1e-12 + np.abs(a))) <= 1e-9, nm
    # ... and alone, rank 0's kind of shard does batch (the agreement, not the planner, held it back)
    alone = cuadmm_amd.SDPSolver(verbose=False)
    alone.init_problem(_amd(make_synthetic([32] * shards[0][3], cons_per_block=3, seed=5)))
    alone.solve(iters, 0.0, 0, 50, 100, 0, 1.05)
    assert alone.counters()["batch_launches"] > 0


@pytest.mark.parametrize("name", ["truss5", "hinf12"])
def test_coupled_constraints_fall_back_to_the_replicated_solve(name, problem_dirs):
    """truss5 (34 blocks, every constraint couples several of them): no constraint is owned by one rank, the general
    path (A*X all-reduced, solve replicated) is taken.  hinf12 (blocks 6, 6, 12): the contiguous split balanced by n^3
    leaves rank 1 without blocks -- and, all constraints being owned by rank 0, without constraints."""
    from oracle import cuadmm_oracle as orc
    q = orc.load_problem_txt(problem_dirs[name])
    p = cuadmm_amd.Problem(q.vec_len, q.con_num, q.blk, q.At_col_ptrs, q.At_row_ids, q.At_vals, q.b_idx, q.b_vals, q.C_idx, q.C_vals)
    ref = cuadmm_amd.SDPSolver(verbose=False)
    ref.init_problem(p)
    ref.solve(30, 0.0, 0, 50, 100, 11000, 1.05)
    world = 2
    ar = HostAllReduce(world)
    solvers = [cuadmm_amd.SDPSolver(verbose=False, rank=r, world=world, profile=1) for r in range(world)]
    errs = []

    def run(r):
        try:
            solvers[r].set_allreduce(ar.hook(r))
            solvers[r].init_problem(p)
            solvers[r].solve(30, 0.0, 0, 50, 100, 11000, 1.05)
        except Exception as e:          # pragma: no cover
            errs.append(e)
            ar.barrier.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for nm in ("errRp", "errRd", "pobj", "dobj", "relgap"):
        a, b0 = ref.info_arr(nm), solvers[0].info_arr(nm)
        assert np.max(np.abs(a - b0) / (1e-12 + np.abs(a))) <= 1e-8, nm
    assert np.max(np.abs(solvers[1].y - ref.y)) <= 1e-8 * (1 + np.max(np.abs(ref.y)))


def test_world_without_hook_is_an_error():
    p = make_synthetic([8] * 6, seed=1)
    s = cuadmm_amd.SDPSolver(verbose=False, rank=0, world=2)
    with pytest.raises(cuadmm_amd.CuadmmError) as e:
        s.init_problem(_amd(p))
    assert e.value.code == -6
