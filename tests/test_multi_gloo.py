"""world_size-2 and -4 tests of the sharded iteration on CPU (gloo): the partition comes from the C ABI
(cuadmm_partition_blocks), each rank iterates only its contiguous block range with the oracle's
arithmetic, the [A*X | sums | A*(S-C)] packet is all-reduced with torch.distributed, the host solve
is replicated.  Result must equal the unsharded oracle run (sharding changes only the
summation order of A*v)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import cuadmm_oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, iters, sw, out):
    import torch
    import torch.distributed as dist
    import cuadmm_amd
    from cuadmm_amd.synthetic import make_synthetic
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = cuadmm_amd.load()
    blk = np.array([6] * 10 + [12] * 7 + [3] * 9 + [20] * 2, dtype=np.int32)
    p = make_synthetic(blk, cons_per_block=2, seed=5)
    first = np.zeros(world + 1, np.int32)
    assert lib.cuadmm_partition_blocks(blk.ctypes.data_as(C.c_void_p), blk.size, world, first.ctypes.data_as(C.c_void_p)) == 0
    off = orc.svec_block_offsets(blk)
    b0, b1 = int(off[first[rank]]), int(off[first[rank + 1]])
    lblk = blk[first[rank]:first[rank + 1]]
    bidx = orc.BlockIndex(lblk)
    # full oracle init gives the scaled data; the shard keeps rows [b0,b1) of At
    full = orc.OracleSolver().init_problem(p)
    At_l = full.At_csr[b0:b1]
    A_l = At_l.T.tocsr()
    C_l, X, S = full.C[b0:b1].copy(), full.X[b0:b1].copy(), full.S[b0:b1].copy()
    y = full.y.copy()
    m = p.con_num

    def reduce(v):
        t = torch.from_numpy(v)
        dist.all_reduce(t)
        return t.numpy()

    ax = reduce(A_l @ X); Rp = full.b - ax
    asmc = reduce(A_l @ (S - C_l))
    sig, errRd = full.sig, full.errRd
    prim_win = dual_win = 0
    traj = []
    for it in range(1, iters + 1):
        y = full._solve(-asmc + Rp / sig)
        Rd1 = At_l @ y - C_l
        Xp = orc.psd_project_svec(bidx, X + sig * Rd1)
        S = (Xp - X) / sig - Rd1
        if it < sw:
            asmc = reduce(A_l @ (S - C_l))
            y = full._solve(-asmc + Rp / sig)
            Rd1 = At_l @ y - C_l
        Rd = Rd1 + S
        tau = 1.95 if it < sw else 1.618
        X = X + tau * sig * Rd
        packet = np.concatenate([A_l @ X, [Rd @ Rd, C_l @ X], A_l @ (S - C_l)])
        packet = reduce(packet)
        Rp = full.b - packet[:m]
        if it >= sw:
            asmc = packet[m + 2:]
        errRp = np.linalg.norm(full.normA * Rp * full.bscale) / full.norm_borg
        errRd = np.sqrt(packet[m]) * full.Cscale / full.norm_Corg
        pobj, dobj = packet[m + 1] * full.objscale, float(full.b @ y) * full.objscale
        if errRp / errRd < 1:
            prim_win += 1
        else:
            dual_win += 1
        if it % 50 == 1:
            if prim_win > 1.2 * dual_win:
                prim_win = 0; sig = min(1e3, sig * 1.05)
            elif dual_win > 1.2 * prim_win:
                dual_win = 0; sig = max(1e-3, sig / 1.05)
        traj.append((errRp, errRd, pobj, dobj, sig))
    np.save(out % rank, np.array(traj))
    np.save((out % rank) + ".X.npy", np.concatenate([[b0, b1], X]))
    dist.destroy_process_group()


@pytest.mark.parametrize("sw,world", [(0, 2), (10 ** 9, 2), (10 ** 9, 4)])
def test_sharded_iteration_matches_unsharded_oracle(tmp_path, sw, world):
    import torch.multiprocessing as mp
    from cuadmm_amd.synthetic import make_synthetic
    iters = 8
    out = str(tmp_path / "traj_%d.npy")
    mp.spawn(_worker, args=(world, _free_port(), iters, sw, out), nprocs=world, join=True)
    blk = np.array([6] * 10 + [12] * 7 + [3] * 9 + [20] * 2, dtype=np.int32)
    p = make_synthetic(blk, cons_per_block=2, seed=5)
    ref = orc.OracleSolver().init_problem(p)
    info = ref.solve(iters, 0.0, 0, 50, 100, sw, 1.05)
    t0 = np.load(out % 0)
    for r in range(1, world):
        assert np.array_equal(t0, np.load(out % r))     # every rank carries the same replicated scalars
    want = np.array([info.errRp, info.errRd, info.pobj, info.dobj, info.sig]).T
    assert np.max(np.abs(t0 - want) / (1e-9 + np.abs(want))) <= 1e-8
    xs = [np.load((out % r) + ".X.npy") for r in range(world)]
    assert xs[0][0] == 0 and xs[-1][1] == p.vec_len and all(xs[r][1] == xs[r + 1][0] for r in range(world - 1))
    Xs = np.concatenate([x[2:] for x in xs]) * ref.bscale
    assert np.max(np.abs(Xs - ref.X)) <= 1e-9 * (1 + np.max(np.abs(ref.X)))
