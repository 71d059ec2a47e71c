#!/usr/bin/env python3
"""Benchmark of the SDP-ADMM iteration hot path on MI355X (contract: see the task brief / DESIGN.md).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): synthetic strictly-feasible SDP with 10 000 PSD blocks of 32x32 PER GPU
(5 constraints/block, 8 nnz each, dense C; cuadmm_amd.synthetic.config_c2), fp64, ADMM-only iterations
(switch_admm=0), stop_tol=0 so exactly K iterations run.  A "step" is one ADMM iteration over one
10 000-block shard: at N GPUs the job is ONE SDP with N*10 000 blocks sharded by block index (weak scaling),
with an RCCL all-reduce of the A*svec(X) partials before each host y-solve; value = N * iterations / time.
Prints one JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BLOCKS_PER_GPU = 10000
BLOCK_N = 32
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
FP64_PEAK_TFLOPS = 78.6        # MI355X FP64 vector = matrix peak (256 CU x 128 FLOP/clk x 2.4 GHz)


def cpu_baseline(prob, n_iters, threads):
    """The oracle (numpy restatement of solver.cu) with the reference's eig_cpu layout: per-block LAPACK
    dsyevd on `threads` host threads over contiguous block ranges (duo_solver.cu:344-371,598-606)."""
    from concurrent.futures import ThreadPoolExecutor
    try:
        from threadpoolctl import threadpool_limits
    except Exception:                                   # pragma: no cover
        threadpool_limits = None
    from oracle import cuadmm_oracle as orc

    n = BLOCK_N
    ii, jj = np.tril_indices(n)
    scale_in = np.where(ii == jj, 1.0, orc.SQRT2INV)
    scale_out = np.where(ii == jj, 1.0, orc.SQRT2)
    seg = n * (n + 1) // 2
    pool = ThreadPoolExecutor(max_workers=threads)

    def chunk(x2d):
        M = np.zeros((x2d.shape[0], n, n))
        v = x2d * scale_in[None, :]
        M[:, jj, ii] = v
        M[:, ii, jj] = v
        w, V = np.linalg.eigh(M)                         # LAPACK dsyevd, the reference's eig_cpu.h:31-51
        P = (V * np.maximum(w, 0.0)[:, None, :]) @ np.swapaxes(V, 1, 2)
        return P[:, jj, ii] * scale_out[None, :]

    def eig_fn(_bidx, xb):
        x2d = xb.reshape(-1, seg)
        bounds = np.linspace(0, x2d.shape[0], threads + 1).astype(int)
        parts = list(pool.map(lambda k: chunk(x2d[bounds[k]:bounds[k + 1]]), range(threads)))
        return np.concatenate(parts).reshape(-1)

    ctx = threadpool_limits(limits=1) if threadpool_limits else None
    try:
        s = orc.OracleSolver(eig_fn=eig_fn).init(prob.vec_len, prob.con_num, prob.At_col_ptrs, prob.At_row_ids,
                                                 prob.At_vals, prob.b_idx, prob.b_vals, prob.C_idx, prob.C_vals, prob.blk)
        s.solve(1, 0.0, 0, 50, 100, 0, 1.05)             # warm-up iteration
        t0 = time.perf_counter()
        s.solve(n_iters, 0.0, 0, 50, 100, 0, 1.05, if_first=False)
        dt = time.perf_counter() - t0
    finally:
        if ctx is not None:
            ctx.__exit__(None, None, None)
        pool.shutdown()
    return n_iters / dt, dt


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary (separate --pmc passes,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); None when no summary is present."""
    best = None
    prof = os.path.join(ROOT, "profiles")
    if os.path.isdir(prof):
        for fn in sorted(os.listdir(prof)):
            if fn.endswith("_pmc_hbm_traffic.json"):
                try:
                    d = json.load(open(os.path.join(prof, fn)))
                except Exception:
                    continue
                for k, v in d.items():
                    if kernel_substr in k:
                        best = v.get("hbm_bytes_per_launch")
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--blocks-per-gpu", type=int, default=BLOCKS_PER_GPU)
    ap.add_argument("--mode", choices=["admm", "sgs"], default="admm")
    ap.add_argument("--comm", choices=["torch", "rccl"], default="torch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=12)       # ~12 s of host work on 30 threads
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world

    # CUADMM_BENCH_FORCE_DIST=1 exercises the torch.distributed/RCCL hook with a single rank (transport check)
    force_dist = os.environ.get("CUADMM_BENCH_FORCE_DIST") == "1"
    dist = None
    torch = None
    if world > 1 or force_dist:
        # import torch BEFORE loading the engine so that both share one HIP runtime (same soname, first one wins)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import cuadmm_amd
    from cuadmm_amd.synthetic import config_c2

    lib = cuadmm_amd.load()
    if lib.cuadmm_device_count() < 1:
        sys.exit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")

    prob = config_c2(args.blocks_per_gpu * world, BLOCK_N)
    solver = cuadmm_amd.SDPSolver(device=local_rank, verbose=False, rank=rank, world=world, profile=2, force_comm=force_dist)

    keep = []
    if world > 1 or force_dist:
        if args.comm == "rccl":
            uid = ctypes.create_string_buffer(128)
            if rank == 0:
                cuadmm_amd._lib.check(lib.cuadmm_rccl_unique_id(uid))
            box = [bytes(uid.raw)]
            dist.broadcast_object_list(box, src=0)
            cuadmm_amd._lib.check(lib.cuadmm_use_rccl(solver._h, box[0], rank, world))
        else:
            class _Ptr:                                   # device pointer -> torch tensor (no copy)
                def __init__(self, ptr, count):
                    self.__cuda_array_interface__ = {"data": (ptr, False), "shape": (count,), "typestr": "<f8", "version": 2}
            cache = {}

            def allreduce(ptr, count, stream):
                key = (ptr, count)
                if key not in cache:
                    cache[key] = torch.as_tensor(_Ptr(ptr, count), device=torch.device("cuda", local_rank))
                ext = torch.cuda.ExternalStream(stream, device=torch.device("cuda", local_rank))
                with torch.cuda.stream(ext):
                    dist.all_reduce(cache[key], op=dist.ReduceOp.SUM)
            solver.set_allreduce(allreduce)
            keep.append(allreduce)

    solver.init_problem(cuadmm_amd.Problem(prob.vec_len, prob.con_num, prob.blk, prob.At_col_ptrs, prob.At_row_ids,
                                           prob.At_vals, prob.b_idx, prob.b_vals, prob.C_idx, prob.C_vals))
    switch = 0 if args.mode == "admm" else 10 ** 9

    def sync():
        cuadmm_amd._lib.check(lib.cuadmm_dev_sync())
        if dist is not None:
            dist.barrier()
            cuadmm_amd._lib.check(lib.cuadmm_dev_sync())

    solver.solve(args.warmup, 0.0, 0, 50, 100, switch, 1.05)
    solver.reset_profile()
    sync()
    t0 = time.perf_counter()
    solver.solve(args.steps, 0.0, 0, 50, 100, switch, 1.05, if_first=False)
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert solver.info_iter_num == args.steps
    prof = solver.profile()
    st = solver.state()

    if rank == 0:
        L_local = args.blocks_per_gpu * BLOCK_N * (BLOCK_N + 1) // 2
        psd = prof["psd_project"]
        psd_ms = psd["ms"] / max(psd["launches"], 1)
        alg_bytes = 16.0 * L_local                         # read Xb + write Xproj, 8 B each per svec element
        achieved = alg_bytes / (psd_ms * 1e-3) / 1e9 if psd_ms > 0 else 0.0
        nominal_flops = (32.0 / 3.0) * args.blocks_per_gpu * BLOCK_N ** 3
        issued_flops = args.blocks_per_gpu * (44 * 48 + 24) * 2048.0     # psd_sign_lds.h: SignWave32, SignPsd schedule
        out = {
            "metric": "ADMM iters/sec, 10k x 32-blk synthetic per GPU (+ PSD-proj TFLOP/s in roofline)",
            "value": world * args.steps / dt,
            "unit": "iters/s (one iteration over a 10k-block shard; N shards advance together)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d PSD blocks of %dx%d per GPU, m=5/blk, nnz=8/constraint, dense C, %s"
                                   % (args.blocks_per_gpu, BLOCK_N, BLOCK_N, "ADMM-only (switch_admm=0)" if args.mode == "admm" else "sGS-ADMM"),
                       "blocks_total": args.blocks_per_gpu * world, "vec_len": int(prob.vec_len), "con_num": int(prob.con_num),
                       "sharding": ("blocks by index; every C2 constraint touches one block, so each rank keeps its own constraints and "
                                    "the ranks all-reduce 4 scalars per iteration (DESIGN.md section 5)") if world > 1 else "single GPU", "comm": args.comm if world > 1 else None},
            # Dominant kernel: psd_sign_wave32_kernel (one wavefront per 32x32 block, matrix-sign iteration on
            # v_mfma_f64_16x16x4_f64).  It is MFMA bound.  `achieved` uses the ALGORITHMIC flops of SURVEY 8d
            # (10.67 n^3 per block, what an eigendecomposition-based projection needs); the flops the kernel really
            # issues on the matrix cores (44 steps x 48 MFMA + 24, 2048 flop each) are reported beside it.
            "roofline": {"kernel": "psd_sign_wave32_kernel (fused svec -> matrix-sign projection -> svec)", "bound": "mfma",
                         "achieved": nominal_flops / (psd_ms * 1e-3) / 1e12 if psd_ms > 0 else 0.0,
                         "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": (nominal_flops / (psd_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS) if psd_ms > 0 else 0.0,
                         "traffic": pmc_traffic("psd_sign_wave32_kernel"), "avg_launch_ms": psd_ms,
                         "algorithmic_flops_per_launch": nominal_flops,
                         "mfma_issued_tflops": issued_flops / (psd_ms * 1e-3) / 1e12 if psd_ms > 0 else 0.0,
                         "mfma_pipe_util": (issued_flops / (psd_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS) if psd_ms > 0 else 0.0,
                         "hbm_gbs": achieved, "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "fp64 matrix-core bound (DESIGN.md section 4); traffic = FETCH_SIZE*2 + WRITE_SIZE from the "
                                 "committed rocprofv3 PMC passes (profiles/), algorithmic bytes 16 B per svec element",
                         "blocks_per_s": args.blocks_per_gpu / (psd_ms * 1e-3) if psd_ms > 0 else 0.0},
            "final_state": {k: st[k] for k in ("errRp", "errRd", "relgap", "sig")},
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = min(30, os.cpu_count() or 1)          # the reference's cpu_eig_thread_num default (main.cu:11)
            v, secs = cpu_baseline(prob, args.cpu_iters, threads)
            out["cpu_baseline"] = {"value": v, "unit": "iters/s", "cores": threads, "kind": "port",
                                   "sample": "%d ADMM iterations of the same 10k x 32 problem on the numpy oracle, "
                                             "per-block LAPACK dsyevd on %d threads (%.1f s)" % (args.cpu_iters, threads, secs)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
