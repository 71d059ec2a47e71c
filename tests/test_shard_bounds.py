"""Host arithmetic of the N-rank paths at world sizes the one-GPU box never runs (the reference's device split
src/duo_solver.cu:269-295, its device walk src/utils/check_gpus.cu:29-43): the row ranges of the sharded dense tail of the
replicated y-solve (cuadmm_tail_shard_bounds = TailSolve::apply's ranges) and the block ranges of the ranks
(cuadmm_partition_blocks) for N = 2 ... 8.  CPU only; the kernels' behaviour on these ranges -- empty ones included -- is
tests/test_gpu_ops.py::test_tail_solve_sharded_partials_sum_to_the_solve."""
import ctypes as C

import numpy as np
import pytest

import cuadmm_amd
from cuadmm_amd._lib import check
from cuadmm_amd.synthetic import config_c4_blk

lib = cuadmm_amd.load()
P = lambda a: a.ctypes.data_as(C.c_void_p)


def bounds(k, world):
    out = np.zeros(world + 1, np.int32)
    check(lib.cuadmm_tail_shard_bounds(k, world, P(out)))
    return out


@pytest.mark.parametrize("k", [50, 1024, 10240, 32768, 17152, 9728, 65536])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 6, 7, 8, 16])
def test_tail_shard_bounds_cover_the_triangle(k, world):
    b = bounds(k, world).astype(np.int64)
    K = (k + 63) // 64 * 64
    assert b[0] == 0 and b[-1] == K                       # [0, K) covered ...
    assert np.all(np.diff(b) >= 0)                        # ... by consecutive ranges (a range may be empty)
    assert np.all(b[:-1] % 8 == 0)                        # the one-pass kernels take rows in groups of 8
    # equal shares of the triangle's ENTRIES (row r has K - r of them): within one group of 8 rows of the ideal share
    entries = (b[1:] - b[:-1]) * K - (b[1:] ** 2 - b[:-1] ** 2) / 2.0
    assert abs(entries.sum() - K * K / 2.0) <= 1e-6 * K * K
    assert np.max(np.abs(entries - K * K / 2.0 / world)) <= 8.0 * K + 64


def test_tail_shard_bounds_small_tail_leaves_ranks_empty():
    """K = 64 on 8 ranks: rows [0, 8) | [8, 8) | [8, 16) ... -- a rank with an empty range still takes part in the all-reduce of the
    partial results with a vector of zeros (checked on the GPU by the sharded tail op)."""
    b = bounds(50, 8)
    assert b[1] == b[2] == 8 and b[-1] == 64
    assert np.count_nonzero(np.diff(b) == 0) >= 1
    with pytest.raises(cuadmm_amd.CuadmmError):
        bounds(0, 2)


@pytest.mark.parametrize("world", [2, 3, 4, 5, 6, 7, 8])
def test_partition_blocks_of_c4_at_every_world_size(world):
    """BASELINE configs[3] (100 000 shuffled blocks of {3, 6, 10, 15, 28, 45}) over N ranks: contiguous ranges in blk order that
    cover every block, cost sum n^3 balanced to within one largest block, svec ranges consistent with the block ranges."""
    blk = np.ascontiguousarray(config_c4_blk(100000), np.int32)
    first = np.zeros(world + 1, np.int32)
    check(lib.cuadmm_partition_blocks(P(blk), blk.size, world, P(first)))
    assert first[0] == 0 and first[-1] == blk.size and np.all(np.diff(first) > 0)
    cost = np.array([np.sum(blk[first[r]:first[r + 1]].astype(np.float64) ** 3) for r in range(world)])
    assert cost.max() - cost.min() <= 2 * 45.0 ** 3
    assert cost.max() <= cost.sum() / world * 1.001


def test_partition_blocks_with_fewer_blocks_than_ranks_and_one_giant():
    """PlanarHand_N=1's census at N = 8 (one block of 120 beside many small ones: the giant holds 13 % of sum n^3, so some ranks get
    nothing but it or nothing at all) and two blocks on eight ranks: every rank gets a valid, possibly EMPTY range."""
    d = np.load(cuadmm_amd.__path__[0] + "/../tests/golden/problems/PlanarHand_N=1_MOMENT.npz")
    blk = np.ascontiguousarray(d["blk"], np.int32)
    for world in (2, 4, 8):
        first = np.zeros(world + 1, np.int32)
        check(lib.cuadmm_partition_blocks(P(blk), blk.size, world, P(first)))
        assert first[0] == 0 and first[-1] == blk.size and np.all(np.diff(first) >= 0)
    first = np.zeros(9, np.int32)
    check(lib.cuadmm_partition_blocks(P(np.array([5, 5], np.int32)), 2, 8, P(first)))
    assert first[-1] == 2 and np.all(np.diff(first) >= 0) and np.count_nonzero(np.diff(first)) == 2
