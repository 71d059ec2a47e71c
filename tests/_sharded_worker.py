"""Worker of tests/test_gpu_sharded_procs.py: one PROCESS per rank (torch.distributed, gloo backend over 127.0.0.1), all
on GPU 0; the all-reduce hook stages the device buffer through the host.  Rank 0 writes the results to argv[1]."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

import cuadmm_amd
from cuadmm_amd._lib import check
from cuadmm_amd.synthetic import make_synthetic

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
lib = cuadmm_amd.load()
coupled = sys.argv[2] == "coupled"
moment = sys.argv[2].split("+")[0].partition(":")[0] in ("pendulum_N=80", "PlanarHand_N=1_MOMENT", "taha1a", "PushBox_N=30_MOMENT")
rng = np.random.default_rng(2)
blk = list(np.array([32] * 20 + [7] * 15 + [15] * 11 + [40, 3, 28])[rng.permutation(49)])
p = make_synthetic(blk, cons_per_block=3, seed=11)
if moment:
    pass
elif coupled:                       # add one constraint over the first entry of every block: no rank owns it
    import scipy.sparse as sp
    off = np.concatenate([[0], np.cumsum(np.array(blk) * (np.array(blk) + 1) // 2)])
    At = sp.csc_matrix((p.At_vals, p.At_row_ids, p.At_col_ptrs), shape=(p.vec_len, p.con_num))
    extra = sp.csc_matrix((np.ones(len(blk)), (off[:-1], np.zeros(len(blk), int))), shape=(p.vec_len, 1))
    At = sp.hstack([At, extra]).tocsc(); At.sort_indices()
    b = np.zeros(p.con_num + 1); b[p.b_idx] = p.b_vals; b[-1] = float(len(blk))
    prob = cuadmm_amd.Problem(p.vec_len, p.con_num + 1, p.blk, At.indptr, At.indices, At.data, np.nonzero(b)[0], b[np.nonzero(b)[0]], p.C_idx, p.C_vals)
else:
    prob = cuadmm_amd.Problem(p.vec_len, p.con_num, p.blk, p.At_col_ptrs, p.At_row_ids, p.At_vals, p.b_idx, p.b_vals, p.C_idx, p.C_vals)


def hook(ptr, count, stream):
    check(lib.cuadmm_dev_sync())
    h = np.empty(count)
    check(lib.cuadmm_memcpy_d2h(h.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), count * 8))
    t = torch.from_numpy(h)
    dist.all_reduce(t)
    check(lib.cuadmm_memcpy_h2d(C.c_void_p(ptr), h.ctypes.data_as(C.c_void_p), count * 8))


if moment:
    # BASELINE configs[4] / [0] sharded: coupled constraints, replicated device-side solve (GPU tail + lead solve) on every rank,
    # 2m+2 all-reduce per half iteration; compared with the committed ORACLE trajectory by the test.  Several cases in one launch
    # ("a+b+c": the ranks' start-up -- eight torch imports on a 16-CPU box -- is paid once); rank 0 writes <argv[1]>.<i>.npz per case.
    from tests.conftest import load_npz_problem
    from tests.helpers import problem_to_amd
    for idx, case in enumerate(sys.argv[2].split("+")):
        name, _, variant = case.partition(":")
        prob = problem_to_amd(load_npz_problem(name))
        # ":hybrid": the host-optimal tail with L21 on the device and the L11 sweeps on the host (lead_solve.h), forced -- on every rank;
        # ":noshard": every rank applies the whole dense tail; ":nonext": the y-solve of iteration k + 1 is NOT enqueued ahead of the
        # host's wait for iteration k (option solve_next = 0: the A/B of that path, same trajectory bit for bit);
        # ":timers": per-class event timers on (profile = 1; they also switch solve_next off)
        opts = {"hybrid": {"tail_k": 10240, "l21_device": 2}, "noshard": {"tail_shard": 0}, "nonext": {"solve_next": 0}}.get(variant)
        calls = [0]

        def counted(ptr, count, stream, calls=calls):
            calls[0] += 1
            hook(ptr, count, stream)
        s = cuadmm_amd.SDPSolver(device=0, verbose=False, rank=rank, world=world, options=opts, profile=1 if variant == "timers" else 0)
        s.set_allreduce(counted)
        s.init_problem(prob)
        s.solve(60, 0.0, 0, 50, 100, 11000, 1.05)
        ti = s.tail_info()
        # every rank's share of the dense tail (bytes of inv(L22) read per solve, bytes resident), gathered for the test
        mine = torch.zeros(world, 3, dtype=torch.float64)
        mine[rank, 0], mine[rank, 1], mine[rank, 2] = ti["bytes_read_per_solve"], ti["rows"], ti["bytes_resident"]
        dist.all_reduce(mine)
        if rank == 0:
            np.savez("%s.%d.npz" % (sys.argv[1], idx), counters=np.array(list(s.counters().values())), shard=np.array(s.shard()),
                     tail_bytes=ti["bytes_read_per_solve"], tail_by_rank=mine.numpy(), allreduce_launches=calls[0], world=world,
                     **{nm: s.info_arr(nm) for nm in ("errRp", "errRd", "pobj", "dobj", "relgap", "sig")})
        del s
        dist.barrier()
    dist.destroy_process_group()
    sys.exit(0)

s = cuadmm_amd.SDPSolver(device=0, verbose=False, rank=rank, world=world)
s.set_allreduce(hook)
s.init_problem(prob)
s.solve(15, 0.0, 0, 50, 100, 8, 1.05)
b0, e0, _, _ = s.shard()
X = np.zeros(prob.vec_len); X[b0:e0] = s.X
Xt = torch.from_numpy(X); dist.all_reduce(Xt)
if rank == 0:
    ref = cuadmm_amd.SDPSolver(device=0, verbose=False)
    ref.init_problem(prob)
    ref.solve(15, 0.0, 0, 50, 100, 8, 1.05)
    np.savez(sys.argv[1], X=X, y=s.y, pobj=s.info_arr("pobj"), errRp=s.info_arr("errRp"), dobj=s.info_arr("dobj"), errRd=s.info_arr("errRd"),
             Xref=ref.X, yref=ref.y, pobj_ref=ref.info_arr("pobj"), errRp_ref=ref.info_arr("errRp"), dobj_ref=ref.info_arr("dobj"),
             errRd_ref=ref.info_arr("errRd"), dims=np.array(s.dims()))
dist.barrier()
dist.destroy_process_group()
