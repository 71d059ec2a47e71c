"""The CPU baseline timed by bench.py (oracle/cpu_eig_baseline.c: the reference's eig_cpu path -- per-block LAPACK dsyevd on
T host threads, include/cuadmm/eig_cpu.h:31-51, src/duo_solver.cu:344-371,598-606) against the numpy oracle."""
import numpy as np
import pytest

from oracle import cpu_baseline as cb
from oracle import cuadmm_oracle as orc


@pytest.mark.parametrize("engine", ["lapack", "ql"])
@pytest.mark.parametrize("threads", [1, 3, 16])
def test_cpu_projection_matches_the_oracle(engine, threads):
    rng = np.random.default_rng(4)
    blk = np.array([1, 2, 5, 32, 32, 17, 45, 3, 64, 10, 28], np.int32)
    L = int(np.sum(blk.astype(np.int64) * (blk + 1) // 2))
    x = rng.standard_normal(L)
    ref = orc.psd_project_svec(orc.BlockIndex(blk), x)
    out, secs = cb.psd_project(x, blk, threads, engine=engine)
    assert secs >= 0
    assert np.max(np.abs(out - ref)) <= 1e-12 * (1 + np.max(np.abs(x)))


def test_thread_split_is_the_references():
    # duo_solver.cu:344-371: floor(n/T) each, the last thread takes the rest, then one-by-one balancing -- every block is
    # projected exactly once whatever T is (more threads than blocks included)
    rng = np.random.default_rng(1)
    blk = np.full(37, 6, np.int32)
    x = rng.standard_normal(37 * 21)
    ref, _ = cb.psd_project(x, blk, 1)
    for t in (2, 5, 30, 64):
        out, _ = cb.psd_project(x, blk, t)
        assert np.array_equal(out, ref)
