"""The transport the north star names -- RCCL -- on the ONE GPU of the test box, at world = 1: both ways the engine reaches it
(cuadmm_use_rccl: a communicator of its own, symbols resolved with dlsym; bench.py's torch.distributed hook on the engine's
stream: backend "nccl" = RCCL) run every collective of a coupled solve, and change nothing: the sum over one rank is the
identity.  (The reference's inter-device exchange: src/duo_solver.cu:487-577, src/utils/check_gpus.cu:29-43.)"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import cuadmm_amd
from cuadmm_amd._lib import check
from tests.conftest import load_npz_problem
from tests.helpers import problem_to_amd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_direct_rccl_communicator_on_a_coupled_problem_against_the_oracle():
    """cuadmm_rccl_unique_id + cuadmm_use_rccl(rank 0 of 1) with option force_comm: pendulum N = 80 (coupled constraints: the 2m+2
    all-reduce before every y-solve, twice per sGS iteration) goes through ncclAllReduce on the engine's stream -- against the committed
    oracle trajectory at the one-rank tolerance, and bit for bit equal to the solve without a communicator."""
    from tests.test_gpu_moment_parity import TOL, SIX, rel_dev
    lib = cuadmm_amd.load()
    with open(os.path.join(ROOT, "tests", "golden", "oracle_traj_moment.json")) as f:
        rec = json.load(f)["pendulum_N=80/switch=11000"]
    prob = problem_to_amd(load_npz_problem("pendulum_N=80"))
    uid = C.create_string_buffer(128)
    check(lib.cuadmm_rccl_unique_id(uid))
    assert any(uid.raw)
    s = cuadmm_amd.SDPSolver(verbose=False, force_comm=True, profile=1)
    check(lib.cuadmm_use_rccl(s._h, uid.raw, 0, 1))
    s.init_problem(prob)
    s.solve(60, 0.0, 0, 50, 100, 11000, 1.05)
    assert s.profile()["allreduce"]["launches"] >= 2 * 60
    plain = cuadmm_amd.SDPSolver(verbose=False, profile=1)
    plain.init_problem(prob)
    plain.solve(60, 0.0, 0, 50, 100, 11000, 1.05)
    assert plain.profile()["allreduce"]["launches"] == 0
    for nm in SIX:
        ref = np.array([float(x) for x in rec[nm]])
        assert rel_dev(s.info_arr(nm)[:ref.size], ref, nm) <= TOL["pendulum_N=80/switch=11000"][0], nm
        assert np.array_equal(s.info_arr(nm), plain.info_arr(nm)), nm
    assert np.array_equal(s.info_arr("sig"), plain.info_arr("sig"))
    assert np.array_equal(s.X, plain.X) and np.array_equal(s.S, plain.S)
    # a second communicator in the same process (bench.py's supplementary solver does this)
    uid2 = C.create_string_buffer(128)
    check(lib.cuadmm_rccl_unique_id(uid2))
    s2 = cuadmm_amd.SDPSolver(verbose=False, force_comm=True)
    check(lib.cuadmm_use_rccl(s2._h, uid2.raw, 0, 1))
    s2.init_problem(prob)
    s2.solve(5, 0.0, 0, 50, 100, 11000, 1.05)
    assert np.array_equal(s2.info_arr("pobj"), plain.info_arr("pobj")[:5])


def _bench(extra, env_extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--no-cpu-baseline", "--blocks-per-gpu", "600", "--steps", "12",
                        "--warmup", "3", "--sharding", "allreduce"] + extra, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_through_rccl_at_world_one():
    """`CUADMM_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --sharding allreduce --comm rccl|torch`: the bench's two transports -- the
    engine's own RCCL communicator, and torch.distributed's (backend nccl = RCCL) on an ExternalStream wrapping the engine's stream --
    with one rank: rc 0, one line, all-reduce launches in the per-class breakdown, and the final state of the timed solve equal bit
    for bit to the run without any transport."""
    plain = _bench([], {})
    assert plain["config"]["comm"] is None and "allreduce" not in plain["breakdown_ms_per_iter"]
    for comm, port in (("rccl", "29671"), ("torch", "29672")):
        d = _bench(["--comm", comm], {"CUADMM_BENCH_FORCE_DIST": "1", "MASTER_PORT": port})
        assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["comm"] == comm
        assert d["breakdown_ms_per_iter"].get("allreduce", 0.0) > 0.0, d["breakdown_ms_per_iter"]
        assert d["final_state"] == plain["final_state"], (comm, d["final_state"], plain["final_state"])
