"""Two PROCESSES (torch.distributed.run, gloo rendezvous on 127.0.0.1) share GPU 0 and run one sharded solve each way:
owned constraints (block-diagonal problem) and the replicated solve (one constraint couples all blocks).  This is the
process / rendezvous structure of `bench.py --gpus N`; only the transport differs (host-staged gloo instead of RCCL)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("kind,port", [("owned", 29641), ("coupled", 29642)])
def test_two_processes_one_gpu(kind, port, tmp_path):
    out = tmp_path / "res.npz"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "_sharded_worker.py"), str(out), kind],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    d = np.load(out)
    assert np.max(np.abs(d["pobj"] - d["pobj_ref"]) / (1e-12 + np.abs(d["pobj_ref"]))) <= 1e-9
    assert np.max(np.abs(d["errRp"] - d["errRp_ref"]) / (1e-12 + np.abs(d["errRp_ref"]))) <= 1e-8
    # the four scalars of the stopping test: summed over ranks on the device in owned-constraints mode (rp_stats_kernel)
    assert np.max(np.abs(d["dobj"] - d["dobj_ref"]) / (1e-12 + np.abs(d["dobj_ref"]))) <= 1e-9
    assert np.max(np.abs(d["errRd"] - d["errRd_ref"]) / (1e-12 + np.abs(d["errRd_ref"]))) <= 1e-8
    # cuadmm_get_dims reports the caller's numbering in every sharding mode (the y buffer is sized from it)
    assert tuple(d["dims"]) == (d["X"].size, d["y"].size, 49)
    assert np.max(np.abs(d["X"] - d["Xref"])) <= 1e-9 * (1 + np.max(np.abs(d["Xref"])))
    assert np.max(np.abs(d["y"] - d["yref"])) <= 1e-8 * (1 + np.max(np.abs(d["yref"])))


def _launch(world, port, cases, out, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "_sharded_worker.py"), str(out), "+".join(cases)],
                       env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    return [np.load("%s.%d.npz" % (out, i)) for i in range(len(cases))]


# ONE launch per world size and group of cases: the ranks' start-up (N torch imports, N HIP contexts on one GPU) is paid once per group
@pytest.mark.parametrize("world,port,cases", [
    (2, 29643, ["pendulum_N=80", "pendulum_N=80:nonext", "pendulum_N=80:timers", "pendulum_N=80:noshard", "PlanarHand_N=1_MOMENT",
                "PlanarHand_N=1_MOMENT:noshard", "taha1a"]),
    (2, 29646, ["PushBox_N=30_MOMENT", "PushBox_N=30_MOMENT:hybrid"]),
    (4, 29647, ["taha1a", "pendulum_N=80"]),
    (8, 29648, ["pendulum_N=80", "PlanarHand_N=1_MOMENT"]),
])
def test_ranks_of_a_moment_relaxation_against_the_oracle(world, port, cases, tmp_path):
    """BASELINE configs[4] (pendulum N = 80) and configs[0] (PlanarHand) on 2, 4 and EIGHT ranks (one process each, all on GPU 0; the
    reference walks devices 0 .. N-1, src/utils/check_gpus.cu:29-43): blocks sharded by index, coupled constraints, the replicated
    y-solve with the GPU tail and the device-side leading sweeps on every rank -- against the committed oracle trajectory
    (tests/golden/oracle_traj_moment.json), same tolerance as the one-rank test."""
    import json
    from tests.test_gpu_moment_parity import TOL, SIX, rel_dev
    with open(os.path.join(ROOT, "tests", "golden", "oracle_traj_moment.json")) as f:
        traj = json.load(f)
    res = dict(zip(cases, _launch(world, port, cases, tmp_path / "res")))
    for name, d in res.items():
        base = name.partition(":")[0]
        rec = traj[base + "/switch=11000"]
        assert int(d["world"]) == world and d["shard"][0] == 0 and 0 < d["shard"][1]     # rank 0 holds a proper part of the svec
        th = TOL[base + "/switch=11000"][0]
        # PushBox_N=30 (m = 154 256) -- the whole y-solve on the device (round 5: a small tail behind dense tree tops, counter 3), and
        # ":hybrid", L21 on the device with the L11 sweeps on the host pool of every rank (2)
        if base.startswith("PushBox"):
            assert d["counters"][6] == (2 if name.endswith(":hybrid") else 3)
        for nm in SIX:
            ref = np.array([float(x) for x in rec[nm]])
            dev = rel_dev(d[nm][:ref.size], ref, nm)
            assert dev <= th, (name, world, nm, dev)
        assert np.array_equal(d["sig"][:60], np.array([float(x) for x in rec["sig"]])), name
        assert d["allreduce_launches"] >= 2 * 60, name               # the collective the north star names ran: twice per sGS iteration
        # The dense tail of the replicated y-solve is split by rows over the ranks (tail_solve.h; the reference's device split
        # src/duo_solver.cu:269-295): a rank reads 1 / world of inv(L22) per solve -- equal shares of the triangle's entries, to within
        # a group of 8 rows -- and the K partial results take one more all-reduce per solve; ":noshard" (option tail_shard = 0) is the
        # replicated tail of rounds 2 - 4
        tk = int(d["counters"][7])
        if tk > 0 and d["counters"][6] in (1, 3):
            K = (tk + 63) // 64 * 64
            whole = 4.0 * K * K
            by_rank = d["tail_by_rank"][:, 0]
            resident = d["tail_by_rank"][:, 2]
            if name.endswith(":noshard"):
                assert np.all(np.abs(by_rank / whole - 1.0) <= 0.01), (name, by_rank / whole)
                assert np.all(resident >= 16.0 * K * K)                  # W and W^T whole on every rank
            else:
                # ... and KEEPS only those rows (round 6, TailSolve::keep_shard): at most twice its share of the triangle (a row is stored up
                # to the range's widest one) + the partial-result vectors, instead of two K x K squares
                assert np.all(resident <= 8.0 * K * K / world + 8.0 * K * 400 + 1e6), (name, world, resident / (16.0 * K * K))
                assert np.all(np.abs(by_rank - whole / world) <= 64.0 * K + 512), (name, world, by_rank / whole)
                assert abs(by_rank.sum() - whole) <= 8.0 * K + 512
                assert d["allreduce_launches"] >= 4 * 60, name        # + one per solve for the partial results
    # the y-solve of iteration k + 1 enqueued ahead of the host's wait (option solve_next, on by default without the per-class timers):
    # the blocking tail_shard all-reduce is then issued from inside fetch_out -- same arithmetic, same order: bit-identical trajectories
    if "pendulum_N=80:nonext" in res:
        for nm in SIX + ("sig",):
            assert np.array_equal(res["pendulum_N=80"][nm], res["pendulum_N=80:nonext"][nm]), nm
            assert np.array_equal(res["pendulum_N=80"][nm], res["pendulum_N=80:timers"][nm]), nm
