#!/usr/bin/env python3
"""Benchmark of the SDP-ADMM iteration hot path on MI355X (contract: see the task brief / DESIGN.md section 6).

    python bench.py --gpus 1 --steps 200 --warmup 20                       # the headline line (BASELINE configs[1])
    python bench.py --gpus N --steps K --warmup W                          # N > 1 as typed: starts its own N ranks (launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                             # the same under a launcher (the driver's form)
    python bench.py --config c1            # PlanarHand_N=1 moment relaxation (BASELINE configs[0]; inputs rebuilt from the shipped .mat)
    python bench.py --config c3            # max-cut, one block n = 2000 (BASELINE configs[2])
    python bench.py --config c4            # 100 000 mixed moment-SOS blocks (BASELINE configs[3]; --scaling strong at N > 1)
    python bench.py --config c5            # pendulum N = 80 trajectory SDP (BASELINE configs[4]); c1 / c5 shard by block index
                                           # over the general path (all-reduce + replicated device-side solve) at N > 1
    ... --c-sparse                         # SURVEY 8d variant of c2: C = A^T y0 + svec(I) (sparse C)
    ... --projection-only                  # SURVEY 8d micro-benchmark: Xb ~ N(0,1)^L through the projection kernels alone
    ... --sharding allreduce               # force the general sharded path: RCCL all-reduce of [A X | sums | A(S-C)] (2m+2
                                           # doubles) before every replicated host y-solve, instead of owned constraints

Workload c2 (default): synthetic strictly-feasible SDP with 10 000 PSD blocks of 32x32 PER GPU (5 constraints/block, 8 nnz
each, dense C; cuadmm_amd.synthetic.config_c2), fp64, ADMM-only iterations (switch_admm=0), stop_tol=0 so exactly K
iterations run.  A "step" is one ADMM iteration over one 10 000-block shard: at N GPUs the job is ONE SDP with N*10 000
blocks sharded by block index (weak scaling); value = N * iterations / time.  The C2 problem is block-diagonal (every
constraint touches one block), so by default each rank keeps its own constraints and the ranks exchange only the four
scalars of the stopping test, folded into one small all-reduce per iteration; `--sharding allreduce`
measures the general path the north star names (A*svec(X) all-reduce + replicated solve) on the same problem.
Prints ONE JSON line on rank 0.
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
SUSTAINED_FP64_MFMA_TFLOPS = 67.9   # measured: every SIMD issuing independent v_mfma_f64_16x16x4_f64 only (tools/ubench/mfma_sustained.hip)


def cpu_baseline(prob, threads, budget_s=20.0):
    """The reference's eig_cpu path beside the GPU number (baseline only, never the target): per-block LAPACK dsyevd on
    `threads` host threads with the reference's static contiguous split (oracle/cpu_eig_baseline.c restating
    include/cuadmm/eig_cpu.h:31-51 and src/duo_solver.cu:344-371,598-606), BLAS threads = 1.  Two numbers on a bounded
    sample: projection-only blocks/s, and whole ADMM iterations/s of the numpy oracle with that projection."""
    from oracle import cpu_baseline as cb
    from oracle import cuadmm_oracle as orc

    blk = np.asarray(prob.blk, np.int32)
    rng = np.random.default_rng(0)
    xb = rng.standard_normal(int(prob.vec_len))
    # warm-up + choice of engine on a slice of the blocks (LAPACK from scipy's OpenBLAS vs the scalar tridiagonal-QL port)
    nslice = max(1, min(blk.size, 8 * threads if blk.max() <= 64 else 1))
    Ls = int(np.sum(blk[:nslice].astype(np.int64) * (blk[:nslice] + 1) // 2))
    # LAPACK leg (the reference's own CPU path: dsyevd('V','U') per block, eig_cpu.h:31-51) at the reference's thread count
    # (cpu_eig_thread_num = 30, main.cu:11) -- the bundled OpenBLAS is built for at most 64 caller threads and aborts beyond
    # ("too many memory regions").  It is the baseline whenever a LAPACK can be dlopen'ed; the build's own scalar
    # Householder + implicit-QL port on every usable core is timed beside it and reported as `port_ql` (it wins on tiny blocks).
    tcount = {"lapack": min(threads, 30), "ql": threads}
    timing = {}
    for eng in ("lapack", "ql"):
        if eng == "ql" and blk.max() > 512:
            continue                                       # the scalar port is far off LAPACK's blocked code at n ~ 2000
        try:
            cb.psd_project(xb[:Ls], blk[:nslice], tcount[eng], engine=eng)
            _, secs = cb.psd_project(xb[:Ls], blk[:nslice], tcount[eng], engine=eng)
        except Exception as exc:                           # no LAPACK on this box: the port alone
            if eng == "lapack":
                timing["lapack_error"] = str(exc)
                continue
            raise
        timing[eng] = secs / nslice
    def timed_projection(e):
        per_block = timing[e]
        n_proj = blk.size if per_block * blk.size <= budget_s / 4 else max(1, int(budget_s / 4 / per_block))
        Lp = int(np.sum(blk[:n_proj].astype(np.int64) * (blk[:n_proj] + 1) // 2))
        _, secs = cb.psd_project(xb[:Lp], blk[:n_proj], tcount[e], engine=e)
        return n_proj, secs

    legs = {}
    for e in ("lapack", "ql"):
        if e in timing:
            n_proj, secs = timed_projection(e)
            legs[e] = {"projection_blocks_per_s": n_proj / secs, "projection_ms": blk.size / (n_proj / secs) * 1e3, "cores": tcount[e],
                       "sample_blocks": n_proj, "sample_s": secs}
    if "lapack" in legs:
        # ONE thread: what a dsyevd('V','U') + DGEMM of these blocks costs without the bundled OpenBLAS's global buffer lock in the way
        # (oracle/cpu_eig_baseline.c: T threads serialise on it for small blocks -- the T-thread rate above is the lock's, not the host's)
        n1 = max(1, min(blk.size, int(1.0 / max(timing["lapack"] * tcount["lapack"], 1e-7))))
        L1 = int(np.sum(blk[:n1].astype(np.int64) * (blk[:n1] + 1) // 2))
        _, s1 = cb.psd_project(xb[:L1], blk[:n1], 1, engine="lapack")
        legs["lapack"]["single_thread_us_per_block"] = s1 / n1 * 1e6
    # `value` = the FASTER engine's projection-bound rate (the rule of rounds 1 - 4: best-of; round 5 led with LAPACK whenever it
    # loaded, which on small blocks is the slower one); the reference's own routine is always in `lapack` beside it
    eng = max(legs, key=lambda e: legs[e]["projection_blocks_per_s"])
    threads_eng = tcount[eng]
    blocks_per_s = legs[eng]["projection_blocks_per_s"]
    proj_s_full = blk.size / blocks_per_s
    n_proj, secs = legs[eng]["sample_blocks"], legs[eng]["sample_s"]

    def eig_fn(_bidx, x):
        return cb.psd_project(x, blk, threads_eng, engine=eng)[0]

    s = orc.OracleSolver(eig_fn=eig_fn).init(prob.vec_len, prob.con_num, prob.At_col_ptrs, prob.At_row_ids,
                                             prob.At_vals, prob.b_idx, prob.b_vals, prob.C_idx, prob.C_vals, prob.blk)
    t0 = time.perf_counter()
    s.solve(1, 0.0, 0, 50, 100, 0, 1.05)                   # first iteration (also measures the per-iteration cost)
    t_it = time.perf_counter() - t0
    n_iters = int(max(1, min(20, (budget_s * 0.5) / max(t_it, 1e-3))))
    t0 = time.perf_counter()
    s.solve(n_iters, 0.0, 0, 50, 100, 0, 1.05, if_first=False)
    dt = time.perf_counter() - t0
    # `value`: the projection-bound rate.  In the reference's eig_cpu mode (src/duo_solver.cu:578-618,793-834) ONLY the eigendecompositions
    # run on the host; every vector stage stays on the GPU, so one iteration costs at least one pass of the host projection (plus the
    # D2H / H2D of the matrices, not charged here).  The numpy oracle's whole iteration (its vector stages are single-threaded numpy,
    # not what the reference would run) is reported beside it as `oracle_iters_per_s`.
    engines = {"lapack": "LAPACK dsyevd('V','U') per block from scipy's bundled OpenBLAS (dlopen, BLAS threads = 1) + DGEMM, static contiguous split "
                         "over the host threads (duo_solver.cu:344-371)",
               "ql": "scalar Householder + implicit-QL port (oracle/eigproj_twin.c, -O3 -march=native), same split"}
    return {"value": 1.0 / proj_s_full, "unit": "iters/s", "cores": threads_eng, "kind": "port",
            "definition": "1 / (host projection of ALL blocks by the faster of the two host engines): the rate the reference's eig_cpu mode is bounded "
                          "by (rounds 1 - 4 reported the numpy oracle's whole iterations with the faster engine -- now `oracle_iters_per_s`; round 5 "
                          "reported this quantity for LAPACK only -- now `lapack`)",
            "eig_engine": "lapack_dsyevd" if eng == "lapack" else "port_ql",
            "nproc": os.cpu_count(), "usable_cpus": usable_cpus(),
            "projection_blocks_per_s": blocks_per_s, "projection_ms": proj_s_full * 1e3,
            "oracle_iters_per_s": n_iters / dt,
            "lapack": dict(legs["lapack"], value=legs["lapack"]["projection_blocks_per_s"] / blk.size, engine=engines["lapack"]) if "lapack" in legs
                      else {"error": timing.get("lapack_error", "")},
            "lapack_single_thread_us_per_block": legs["lapack"]["single_thread_us_per_block"] if "lapack" in legs else None,
            "port_ql": dict(legs["ql"], value=legs["ql"]["projection_blocks_per_s"] / blk.size, engine=engines["ql"]) if "ql" in legs else None,
            "engine": engines[eng],
            "sample": "projection-only: %d of %d blocks in %.2f s on %d host threads -> value = 1 / (host projection of all blocks); "
                      "oracle_iters_per_s: %d whole ADMM iterations of the numpy oracle with that projection, %.1f s" % (
                          n_proj, blk.size, secs, threads_eng, n_iters, dt)}


def _as_synth(p):
    """cuadmm_amd.Problem -> the attribute names of synthetic.SyntheticProblem (what the rest of this file reads)"""
    from cuadmm_amd.synthetic import SyntheticProblem
    return SyntheticProblem(p.vec_len, p.con_num, p.blk_vals, p.At_csc_col_ptrs, p.At_csc_row_ids, p.At_csc_vals, p.b_indices, p.b_vals,
                            p.C_indices, p.C_vals)


def usable_cpus():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU box shows nproc = 256 under
    a 16-CPU quota; 256 busy threads there are 16 CPUs' worth of work)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    return max(1, n)


def pmc_traffic(kernel_substr, config):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary of this configuration
    (profiles/rNN_<config>_pmc_hbm_traffic.json, latest round; separate --pmc passes, FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950); None when no summary is present."""
    best = None
    prof = os.path.join(ROOT, "profiles")
    if os.path.isdir(prof):
        for fn in sorted(os.listdir(prof)):
            if fn.endswith("_%s_pmc_hbm_traffic.json" % config):
                try:
                    d = json.load(open(os.path.join(prof, fn)))
                except Exception:
                    continue
                for k, v in d.items():
                    if kernel_substr in k:
                        # a kernel that runs several iterations per launch is recorded per ITERATION too (the profiled run's launches
                        # need not be as long as this run's): scaled to this run's launch length by the caller
                        best = (v.get("hbm_bytes_per_launch"), v.get("hbm_bytes_per_iteration"))
    return best


def issued_mfma_flops(blk, steps):
    """fp64 flops the projection kernels issue on the matrix cores for the measured per-block step counts
    (v_mfma_f64_16x16x4_f64 = 2048 flop), one-wavefront kernels of psd_sign_wave.h: NT (NT + 1) / 2 upper sub-tiles x 4 NT
    MFMA per product, two products per step + the final one (n <= 16: 8 per step + 4; n <= 32: 48 + 24; n <= 48: 144 + 72;
    n <= 64: 320 + 160); larger: upper-triangle tiles of the batched GEMMs (psd_large.hip).  Blocks served by the register
    eigensolver (n <= 8) issue 0 here (their rebuild is a few MFMAs)."""
    blk = np.asarray(blk, np.int64)
    steps = np.asarray(steps, np.float64)
    fl = np.zeros(blk.size)
    m = (blk <= 16) & (steps > 0)                            # n <= 8 only when they run the sign kernel (closed plans)
    fl[m] = (steps[m] * 8 + 4) * 2048.0
    m = (blk > 16) & (blk <= 32)
    fl[m] = (steps[m] * 48 + 24) * 2048.0
    m = (blk > 32) & (blk <= 48)
    fl[m] = (steps[m] * 2 + 1) * 6 * 12 * 2048.0
    m = (blk > 48) & (blk <= 64)
    fl[m] = (steps[m] * 2 + 1) * 10 * 16 * 2048.0
    m = blk > 64
    if m.any():
        N = (blk[m] + 63) // 64 * 64
        tm = np.where((N >= 128) & (N <= 3000), 32, 64)      # lg_small_tiles (single large matrix / small groups)
        N = np.where((tm == 32) & (blk[m] <= N - 32), N - 32, N)   # 32 x 32 tiles: the padding is a multiple of 32 (SignPsd::build)
        nb = N // tm
        fl[m] = (steps[m] * 2 + 1) * 2.0 * N * tm * tm * (nb * (nb + 1) // 2)
    return float(fl.sum())


def projection_only(args, lib):
    """SURVEY 8d micro-benchmark: Xb ~ N(0,1)^L through the projection kernels alone (cuadmm_op_psd_project: svec in, projected
    svec out, 16 B per svec element), `steps` repetitions after `warmup`."""
    import ctypes as C
    from cuadmm_amd import synthetic
    from cuadmm_amd._lib import check
    from cuadmm_amd.devbuf import Dev
    blk = {"c2": np.full(args.blocks_per_gpu, BLOCK_N), "c3": np.array([2000]), "c4": synthetic.config_c4_blk(args.blocks_per_gpu)}.get(args.config)
    if blk is None:
        d = np.load(os.path.join(ROOT, "tests", "golden", "problems", {"c1": "PlanarHand_N=1_MOMENT", "c5": "pendulum_N=80"}[args.config] + ".npz"))
        blk = d["blk"]
    blk = np.ascontiguousarray(blk, np.int32)
    L = int(np.sum(blk.astype(np.int64) * (blk + 1) // 2))
    x = np.random.default_rng(20240601).standard_normal(L)
    din, dout = Dev(x), Dev(shape=(L,), dtype=np.float64)
    steps_dev = Dev(np.zeros(blk.size, np.int32))
    plan = C.c_void_p()
    check(lib.cuadmm_psd_plan_create(blk.ctypes.data_as(C.c_void_p), int(blk.size), 0, C.byref(plan)))
    for _ in range(max(args.warmup, 1)):
        check(lib.cuadmm_psd_plan_project(plan, din.ptr, dout.ptr, steps_dev.ptr, None))
    check(lib.cuadmm_dev_sync())
    t0 = time.perf_counter()
    for _ in range(args.steps):
        check(lib.cuadmm_psd_plan_project(plan, din.ptr, dout.ptr, steps_dev.ptr, None))
    check(lib.cuadmm_dev_sync())
    dt = (time.perf_counter() - t0) / args.steps
    steps_blk = steps_dev.get()
    nominal = (32.0 / 3.0) * float(np.sum(blk.astype(np.float64) ** 3))
    issued = issued_mfma_flops(blk, steps_blk)
    print(json.dumps({
        "metric": "PSD projections/sec of the %s block mix (projection-only micro-benchmark, SURVEY 8d)" % args.config,
        "value": 1.0 / dt, "unit": "projections/s (one projection of all %d blocks)" % blk.size, "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "Xb ~ N(0,1)^L, blocks of %s, svec in -> projected svec out (cuadmm_psd_plan_project on a plan built once)" % args.config,
                   "blocks_total": int(blk.size), "vec_len": L},
        "roofline": {"kernel": "projection kernels of the size classes present (unfused)", "bound": "mfma", "achieved": nominal / dt / 1e12, "peak": FP64_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": nominal / dt / 1e12 / FP64_PEAK_TFLOPS, "traffic": None, "avg_launch_ms": dt * 1e3,
                     "mfma_issued_tflops": issued / dt / 1e12, "hbm_gbs": 16.0 * L / dt / 1e9, "algorithmic_bytes_per_launch": 16.0 * L,
                     "newton_schulz_steps": {"mean": float(steps_blk[steps_blk > 0].mean()) if (steps_blk > 0).any() else 0.0},
                     "blocks_per_s": blk.size / dt}}), flush=True)
    lib.cuadmm_psd_plan_destroy(plan)


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N rank processes of this script (one per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment -- what torch.distributed.run would set), relay rank 0's JSON line, return the
    worst exit code.  The counterpart of the reference's one-process multi-device launch (src/duo_solver.cu:487-577,
    src/utils/check_gpus.cu:29-43).  The caller has not initialised the GPU: the ranks are children, nothing is exec'ed."""
    import socket
    import subprocess

    port = os.environ.get("MASTER_PORT")
    if port is None:
        with socket.socket() as s:                          # a free port for the rendezvous
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    import tempfile
    procs = []
    with tempfile.TemporaryFile(mode="w+") as out0:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        # a rank that dies takes the job with it (the others would wait in a collective for ever)
        rcs = [None] * n
        while any(rc is None for rc in rcs):
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    rcs[r] = p.poll()
            if any(rc not in (None, 0) for rc in rcs):
                time.sleep(2.0)                             # let the others report before they are stopped
                for r, p in enumerate(procs):
                    if p.poll() is None:
                        p.kill()
                    rcs[r] = p.wait()
                break
            time.sleep(0.05)
        out0.seek(0)
        for ln in out0.read().splitlines():                 # stdout carries the ONE JSON line; library chatter of the ranks goes to stderr
            (sys.stdout if ln.startswith("{") else sys.stderr).write(ln + "\n")
        sys.stdout.flush()
    bad = [rc for rc in rcs if rc != 0]
    if bad:
        sys.stderr.write("bench.py: rank exit codes %s\n" % rcs)
        return max(abs(rc) for rc in bad) or 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", choices=["c1", "c2", "c3", "c4", "c5"], default="c2")
    ap.add_argument("--c-sparse", action="store_true", help="c2 / c4 with the sparse C of SURVEY 8d (C = A^T y0 + svec(I))")
    ap.add_argument("--projection-only", action="store_true", help="time the PSD projection of Xb ~ N(0,1)^L alone (cuadmm_op_psd_project)")
    ap.add_argument("--time-to-tol", type=float, default=None, help="also run a fresh solve to this tolerance (untimed region) and report it")
    ap.add_argument("--convergence-cap", type=int, default=None, help="c5: iteration cap of the convergence run with the reference log's parameters "
                    "(default 100 000, the reference's own; 1 000 000 for profiles/r06_pendulum_convergence.json)")
    ap.add_argument("--blocks-per-gpu", type=int, default=None)
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None)
    ap.add_argument("--sharding", choices=["owned", "allreduce"], default="owned")
    ap.add_argument("--mode", choices=["admm", "sgs"], default="admm")
    ap.add_argument("--comm", choices=["torch", "rccl"], default="torch")
    ap.add_argument("--batch", type=int, default=None, help="ADMM iterations per launch on closed blocks (engine option 'batch'; 0 = one launch per iteration)")
    ap.add_argument("--option", action="append", help="engine option key=value (cuadmm_set_option), repeatable: A/B runs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--no-allreduce-path", action="store_true", help="N > 1, c2 / c4: skip the supplementary run of the general sharded path")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    args = ap.parse_args()
    if args.steps is None:
        args.steps = {"c1": 100, "c2": 200, "c3": 50, "c4": 100, "c5": 200}[args.config]
    if args.warmup is None:
        args.warmup = {"c1": 10, "c2": 20, "c3": 5, "c4": 10, "c5": 20}[args.config]
    if args.scaling is None:
        args.scaling = "weak" if args.config == "c2" else "strong"
    if args.blocks_per_gpu is None:
        args.blocks_per_gpu = {"c1": 122, "c2": BLOCKS_PER_GPU, "c3": 1, "c4": 100000, "c5": 239}[args.config]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # `python bench.py --gpus N` as typed: this process becomes the launcher.  It has not touched the GPU, torch or the
            # engine (a process that has may not be replaced, and a forked HIP runtime is unusable): N fresh rank processes.
            sys.exit(launch_ranks(args.gpus))
        args.gpus = world

    # CUADMM_BENCH_FORCE_DIST=1 exercises the torch.distributed/RCCL hook with a single rank (transport check)
    force_dist = os.environ.get("CUADMM_BENCH_FORCE_DIST") == "1"
    replicas = args.config == "c3"                          # one block: nothing to shard -- N independent replicas
    dist = None
    torch = None
    if world > 1 or force_dist:
        # import torch BEFORE loading the engine so that both share one HIP runtime (same soname, first one wins)
        import torch
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ.setdefault("MASTER_PORT", "29533")
        # CUADMM_BENCH_BACKEND=gloo (tests only): the same script with N ranks SHARING the GPUs that exist (rank r on device
        # r mod count) and a host-staged all-reduce -- what tests/test_gpu_bench_ranks.py runs on a one-GPU box
        backend = os.environ.get("CUADMM_BENCH_BACKEND", "nccl")
        if backend == "gloo":
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            if local_rank >= torch.cuda.device_count():
                sys.exit("bench.py: rank %d of %d has no GPU of its own (%d visible) -- one process per GPU; CUADMM_BENCH_BACKEND=gloo lets the "
                         "ranks share the GPUs that exist (tests)" % (rank, world, torch.cuda.device_count()))
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import cuadmm_amd
    from cuadmm_amd import synthetic

    lib = cuadmm_amd.load()
    if lib.cuadmm_device_count() < 1:
        sys.exit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")

    if args.projection_only:
        return projection_only(args, lib)

    n_units = args.blocks_per_gpu * (world if args.scaling == "weak" else 1)
    if args.config == "c2":
        prob = synthetic.make_synthetic([BLOCK_N] * n_units, dense_C=not args.c_sparse)
        workload = "BASELINE configs[1]: %d PSD blocks of %dx%d %s, m=5/blk, nnz=8/constraint, %s C" % (
            args.blocks_per_gpu, BLOCK_N, BLOCK_N, "per GPU" if args.scaling == "weak" else "in total", "sparse" if args.c_sparse else "dense")
    elif args.config == "c4":
        prob = synthetic.config_c4(n_units) if not args.c_sparse else synthetic.make_synthetic(synthetic.config_c4_blk(n_units), cons_per_block=3, dense_C=False)
        workload = "BASELINE configs[3]: %d moment-SOS blocks of sizes {3,6,10,15,28,45} %s, m=3/blk" % (
            args.blocks_per_gpu, "per GPU" if args.scaling == "weak" else "in total")
    elif args.config in ("c1", "c5"):
        # real data: inputs rebuilt from the reference's shipped .mat files (tests/golden/make_golden.py), committed as fixtures
        name = {"c1": "PlanarHand_N=1_MOMENT", "c5": "pendulum_N=80"}[args.config]
        d = np.load(os.path.join(ROOT, "tests", "golden", "problems", name + ".npz"))
        prob = cuadmm_amd.Problem.from_coo(d["blk"], int(d["con_num"]), d["At_row"], d["At_col"], d["At_val"], d["b_idx"], d["b_val"], d["C_idx"], d["C_val"])
        prob = _as_synth(prob)
        args.scaling = "strong"
        workload = {"c1": "BASELINE configs[0]: examples/SPOT PlanarHand_N=1_MOMENT (122 blocks, n <= 120, m = 66 008; moment relaxation: coupled "
                          "constraints, GPU tail + device-side sweeps of the A A^T solve)",
                    "c5": "BASELINE configs[4]: examples/pendulum N=80 (159 x 10 + 80 x 55, m = 112 028; the largest horizon the reference ships)"}[args.config]
    else:
        prob = synthetic.config_c3(2000)
        workload = "BASELINE configs[2]: max-cut relaxation of G(2000, p = 0.01) (SURVEY 8d), one PSD block n=2000 (N independent replicas at N GPUs)"
    workload += ", " + ("ADMM-only (switch_admm=0)" if args.mode == "admm" else "sGS-ADMM")
    eng_world, eng_rank = (1, 0) if replicas else (world, rank)
    engine_options = {}
    if args.batch is not None:
        engine_options["batch"] = args.batch
    if args.sharding == "allreduce":
        engine_options["local_constraints"] = 0             # keep every constraint on every rank (the general sharded path)
    for kv in args.option or []:
        k, v = kv.split("=")
        engine_options[k] = float(v)
    use_comm = (world > 1 and not replicas) or force_dist
    solver = cuadmm_amd.SDPSolver(device=local_rank, verbose=False, rank=eng_rank, world=eng_world, profile=2,
                                  force_comm=force_dist, psd_steps=True,
                                  options=engine_options)

    keep = []
    if use_comm:
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

            def allreduce_host(ptr, count, stream):     # gloo: through the host (tests)
                cuadmm_amd._lib.check(lib.cuadmm_dev_sync())
                h = np.empty(count)
                cuadmm_amd._lib.check(lib.cuadmm_memcpy_d2h(h.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), count * 8))
                dist.all_reduce(torch.from_numpy(h))
                cuadmm_amd._lib.check(lib.cuadmm_memcpy_h2d(ctypes.c_void_p(ptr), h.ctypes.data_as(ctypes.c_void_p), count * 8))

            def allreduce(ptr, count, stream):
                key = (ptr, count)
                if key not in cache:
                    cache[key] = torch.as_tensor(_Ptr(ptr, count), device=torch.device("cuda", local_rank))
                ext = torch.cuda.ExternalStream(stream, device=torch.device("cuda", local_rank))
                with torch.cuda.stream(ext):
                    dist.all_reduce(cache[key], op=dist.ReduceOp.SUM)
            if dist.get_backend() == "gloo":
                allreduce = allreduce_host
            solver.set_allreduce(allreduce)
            keep.append(allreduce)

    t_init = time.perf_counter()
    solver.init_problem(cuadmm_amd.Problem(prob.vec_len, prob.con_num, prob.blk, prob.At_col_ptrs, prob.At_row_ids,
                                           prob.At_vals, prob.b_idx, prob.b_vals, prob.C_idx, prob.C_vals))
    t_init = time.perf_counter() - t_init
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
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert solver.info_iter_num == args.steps
    prof = solver.profile()
    st = solver.state()
    steps_blk = solver.psd_steps()
    # Supplementary, NOT `value`: the same solver continued for a longer stretch.  The chip ramps its clocks over the first ~30 ms
    # of dense work (tools/probe_rampup.py), so a 20-step timed region right after init sits below the rate a
    # solve of thousands of iterations sees; both numbers are in the line.
    ss_steps = {"c1": 300, "c2": 400, "c3": 40, "c4": 150, "c5": 400}[args.config]
    sync()
    t0 = time.perf_counter()
    solver.solve(ss_steps, 0.0, 0, 50, 100, switch, 1.05, if_first=False)
    sync()
    dt_ss = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt_ss], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_ss = float(t.item())
    # Supplementary, NOT `value`: the timed region runs with stop_tol = 0 (exactly K iterations), so the engine takes no checkpoint
    # before a launch of several iterations; a solve WITH a tolerance checkpoints X, S, y, [A X | A (S - C)] at the start of every
    # batch (one copy kernel).  The same solver continued with a tolerance that is never met: the rate production solves see.
    ck = None
    if args.config in ("c2", "c4") and not args.no_breakdown:
        ck_steps = ss_steps
        sync()
        t0 = time.perf_counter()
        solver.solve(ck_steps, 1e-300, 0, 50, 100, switch, 1.05, if_first=False)
        sync()
        dt_ck = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt_ck], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_ck = float(t.item())
        ck = (ck_steps, dt_ck, solver.info_iter_num)
    # Supplementary, NOT `value`: the boundary takes host arrays in init and hands X, y, S back to host arrays -- the PCIe-inclusive
    # rate of the whole job (init with its uploads + the timed iterations + the read-back of the results), DESIGN.md section 6
    sync()
    t0 = time.perf_counter()
    _x, _y, _s = solver.X, solver.y, solver.S
    sync()
    t_readback = time.perf_counter() - t0
    del _x, _y, _s
    _, _, kb, ke = solver.shard()
    blk_local = np.asarray(prob.blk)[kb:ke]

    # per-class breakdown (hipEvent pairs around every kernel class + host timers), outside the timed region
    breakdown = None
    if not args.no_breakdown:
        cuadmm_amd._lib.check(lib.cuadmm_set_option(solver._h, b"profile", 1.0))
        solver.reset_profile()
        nb = max(5, min(50, args.steps))
        solver.solve(nb, 0.0, 0, 50, 100, switch, 1.05, if_first=False)
        sync()
        breakdown = {k: v["ms"] / nb for k, v in solver.profile().items() if v["launches"]}

    # Supplementary at N > 1 on the block-diagonal configurations: `value` is the owned-constraints path (each rank keeps the
    # constraints of its own blocks: 4 scalars all-reduced per iteration); the collective the north star names -- [A X | sums |
    # A (S - C)] (2m+2 doubles) all-reduced before every replicated y-solve -- runs here on the SAME problem in the same process
    # group (engine option local_constraints = 0), so that one line carries both.
    ar_path = None
    if use_comm and not replicas and world > 1 and args.sharding == "owned" and args.config in ("c2", "c4") and not args.no_allreduce_path:
        opts2 = dict(engine_options)
        opts2["local_constraints"] = 0
        s3 = cuadmm_amd.SDPSolver(device=local_rank, verbose=False, rank=eng_rank, world=eng_world, profile=2, force_comm=force_dist, options=opts2)
        if args.comm == "rccl":
            uid = ctypes.create_string_buffer(128)
            if rank == 0:
                cuadmm_amd._lib.check(lib.cuadmm_rccl_unique_id(uid))
            box = [bytes(uid.raw)]
            dist.broadcast_object_list(box, src=0)
            cuadmm_amd._lib.check(lib.cuadmm_use_rccl(s3._h, box[0], rank, world))
        else:
            s3.set_allreduce(keep[0])
        s3.init_problem(cuadmm_amd.Problem(prob.vec_len, prob.con_num, prob.blk, prob.At_col_ptrs, prob.At_row_ids,
                                           prob.At_vals, prob.b_idx, prob.b_vals, prob.C_idx, prob.C_vals))
        s3.solve(args.warmup, 0.0, 0, 50, 100, switch, 1.05)
        sync()
        t0 = time.perf_counter()
        s3.solve(args.steps, 0.0, 0, 50, 100, switch, 1.05, if_first=False)
        sync()
        dt3 = time.perf_counter() - t0
        t = torch.tensor([dt3], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt3 = float(t.item())
        cuadmm_amd._lib.check(lib.cuadmm_set_option(s3._h, b"profile", 1.0))
        s3.reset_profile()
        nb3 = max(5, min(50, args.steps))
        s3.solve(nb3, 0.0, 0, 50, 100, switch, 1.05, if_first=False)
        sync()
        bd3 = {k: v["ms"] / nb3 for k, v in s3.profile().items() if v["launches"]}
        ar_path = {"value": (world if args.scaling == "weak" else 1) * args.steps / dt3, "ms_per_step": dt3 / args.steps * 1e3,
                   "allreduce_ms_per_iter": bd3.get("allreduce", 0.0), "allreduce_doubles": 2 * int(prob.con_num) + 2,
                   "breakdown_ms_per_iter": bd3,
                   "sharding": "blocks by index; all-reduce of [A X | sums | A(S-C)] (2m+2 doubles) before every replicated y-solve "
                               "(engine option local_constraints = 0), same problem, same ranks",
                   "note": "supplementary: the general sharded path of SURVEY 8e on the configuration whose `value` takes the owned-constraints shortcut"}
        del s3

    time_to_tol = None
    if args.time_to_tol is None and args.config in ("c1", "c5"):
        args.time_to_tol = 1e-3
    if args.time_to_tol and args.config == "c5":
        # BASELINE configs[4] AS WRITTEN ("stop_tol=1e-6, end-to-end convergence"), with the parameters of the reference's own run
        # (examples/pendulum/N=80_licols.log: sGS-ADMM, switch to ADMM at 11 000, sig_update_threshold = 0, stop_tol 1e-6 -- never met in
        # its 100 000 iterations: final relgap 2.7e-4 after 2 218.7 s).  ONE fresh solve with those parameters up to the cap (default: the
        # reference's own 100 000; --convergence-cap 1000000 for profiles/r06_pendulum_convergence.json); the first iteration at which
        # max(errRp, errRd, relgap) falls below 1e-3 / 1e-4 / 1e-5 is read off the per-iteration arrays (the tau rule looks at stop_tol, so
        # separate solves per tolerance would be other trajectories), its time is that fraction of the solve (the iteration cost is flat).
        s2 = cuadmm_amd.SDPSolver(device=local_rank, verbose=False, rank=eng_rank, world=eng_world, force_comm=force_dist, options=engine_options)
        if use_comm:
            s2.set_allreduce(keep[0]) if keep else None
        s2.init_problem(cuadmm_amd.Problem(prob.vec_len, prob.con_num, prob.blk, prob.At_col_ptrs, prob.At_row_ids,
                                           prob.At_vals, prob.b_idx, prob.b_vals, prob.C_idx, prob.C_vals))
        cap = int(args.convergence_cap or 100000)
        sync()
        t0 = time.perf_counter()
        s2.solve(cap, 1e-6, 0, 50, 100, 11000, 1.05)
        sync()
        t1 = time.perf_counter() - t0
        n_it = s2.info_iter_num
        arrs = {k: s2.info_arr(k) for k in ("errRp", "errRd", "relgap", "pobj", "dobj", "sig")}
        kkt = np.maximum(np.maximum(arrs["errRp"], arrs["errRd"]), arrs["relgap"])[:n_it]
        first = {}
        for tol in (1e-3, 1e-4, 1e-5, 1e-6):
            hit = np.nonzero(kkt < tol)[0]
            above = np.nonzero(kkt >= tol)[0]
            # `iteration`: where the reference's stopping test (pointwise: max(errRp, errRd, relgap) < tol) would have ended the solve;
            # `stays_below_from`: the iteration after the LAST one above the tolerance within the cap (None: still above at the cap)
            first["%.0e" % tol] = None if hit.size == 0 else {
                "iteration": int(hit[0]) + 1, "seconds": t1 * (int(hit[0]) + 1) / n_it,
                "stays_below_from": None if (above.size and above[-1] == n_it - 1) else (int(above[-1]) + 2 if above.size else 1)}
        at = {}
        for it in (11000, 100000, 300000, 1000000):
            if it <= n_it:
                at[str(it)] = {k: float(arrs[k][it - 1]) for k in arrs}
        st2 = s2.state()
        time_to_tol = {"parameters": "sGS-ADMM, switch_admm = 11000, sig_update_threshold = 0, stages 50 / 100, sigscale 1.05, stop_tol = 1e-6 "
                                     "(examples/pendulum/N=80_licols.log)",
                       "iteration_cap": cap, "iterations": n_it, "seconds": t1, "ms_per_iteration": t1 / max(n_it, 1) * 1e3,
                       "first_below": first, "residuals_at": at,
                       "final": {k: st2[k] for k in ("errRp", "errRd", "relgap", "pobj", "dobj", "best_KKT")},
                       "reference": {"iterations": 100000, "seconds": 2218.7, "final": {"errRp": 6.3e-05, "errRd": 2.8e-05, "relgap": 2.7e-04},
                                     "best_KKT_after_switch": 1.5e-04, "source": "examples/pendulum/N=80_licols.log (hardware unnamed)"},
                       "note": "stop_tol = 1e-6 is not met by the reference in its 100 000 iterations either; the cap is %s" % (
                           "the reference's own" if cap == 100000 else "%d iterations (--convergence-cap)" % cap)}
        del s2
    elif args.time_to_tol:
        # a fresh solve to the tolerance with the parameters of the reference's logs (untimed region of this benchmark)
        s2 = cuadmm_amd.SDPSolver(device=local_rank, verbose=False, rank=eng_rank, world=eng_world, force_comm=force_dist,
                                  options=engine_options)
        if use_comm:
            s2.set_allreduce(keep[0]) if keep else None
        s2.init_problem(cuadmm_amd.Problem(prob.vec_len, prob.con_num, prob.blk, prob.At_col_ptrs, prob.At_row_ids,
                                           prob.At_vals, prob.b_idx, prob.b_vals, prob.C_idx, prob.C_vals))
        cap = 30000
        sync()
        t0 = time.perf_counter()
        s2.solve(cap, args.time_to_tol, 0, 50, 100, 0 if args.mode == "admm" else 11000, 1.05)
        sync()
        t1 = time.perf_counter() - t0
        st2 = s2.state()
        time_to_tol = {"tol": args.time_to_tol, "iterations": s2.info_iter_num, "seconds": t1, "reached": bool(max(st2["errRp"], st2["errRd"], st2["relgap"]) < args.time_to_tol),
                       "final": {k: st2[k] for k in ("errRp", "errRd", "relgap", "pobj", "dobj")}, "iteration_cap": cap}
        del s2

    if rank == 0:
        psd = prof["psd_project"]
        launches = max(psd["launches"], 1)
        iters_per_launch = args.steps / launches          # > 1 when the engine runs several iterations per launch (option "batch")
        psd_ms = psd["ms"] / args.steps                   # projection time per ITERATION; a launch takes psd_ms * iters_per_launch
        L_local = int(np.sum(blk_local.astype(np.int64) * (blk_local + 1) // 2))
        # read Xb + write Xproj, 8 B each per svec element; blocks of the FUSED iteration (9 <= n <= 64, psd_fuse.h)
        # also carry the vector work of aty_xb / post: row pointer 4 + C 8 + X 8 read, Rd1 8 write | X 8 + Rd1 8 + C 8 read,
        # S 8 + X 8 write = 68 B per svec element, and Xb / Xproj never reach HBM
        plan = solver.counters()
        fused = plan["fused"] > 0
        closed = plan["closed_blocks"] > 0
        fb = blk_local[((blk_local > 8) | closed) & (blk_local <= 64)].astype(np.int64)
        L_fused = int(np.sum(fb * (fb + 1) // 2)) if fused else 0
        # closed blocks (psd_sign_closed.h): X and C read, S and X written (32 B per svec element; Rd1, Xb, Xproj never reach HBM,
        # the kernel re-reads X twice and C once through L2) + one 1.5 KB record per block; fused, not closed: 68 B; else 16 B
        alg_bytes = (32.0 if closed else 68.0) * L_fused + 16.0 * (L_local - L_fused) + (1472.0 * fb.size if closed else 0.0)
        nominal_flops = (32.0 / 3.0) * float(np.sum(blk_local.astype(np.float64) ** 3))
        issued_flops = issued_mfma_flops(blk_local, steps_blk)
        sign_blocks = blk_local > (0 if closed else 8)
        batched = plan["batch_launches"] > 0
        kernel = {"c1": "psd_project phase: lg_sign_cluster_kernel (nine blocks of n = 66 ... 120: the whole matrix-sign iteration in one launch of upper-triangle fp64-MFMA tiles) | psd_sign_lds_kernel (n = 33 ... 64) | psd_sign_wave_kernel (n <= 32) | psd_small_reg_kernel (n <= 8), concurrent streams",
                  "c2": ("psd_sign_closed_cu_kernel<2, 16, 4> (several ADMM iterations of every block per launch: y-solve, A^T y, Xb, adaptive matrix-sign projection, S / X updates, "
                         "A X, A (S - C), partial sums; one persistent workgroup per CU)" if batched else
                         "psd_sign_closed_kernel<2, 4> (whole ADMM iteration of a block, one wavefront per block)" if closed else
                         "psd_sign_wave_kernel<2, 4, fused> (X / A^T y / C -> adaptive matrix-sign projection -> S, X updates; one wavefront per block)"),
                  "c3": "lg_gemm_sym_kernel (adaptive matrix-sign projection of the n=2000 block as batched upper-triangle fp64-MFMA GEMMs)",
                  "c4": "psd_project phase: psd_sign_closed_kernel<3, 2> (n=45) | <2, 4> (n=28) | <1, 8> (n <= 15), whole iteration per block, concurrent streams" if closed else
                        "psd_project phase: psd_sign_wave_kernel<3, 2> (n=45) | <2, 4> (n=28) | <1, 8> (n=10, 15), fused | psd_small_reg_kernel (n<=6), concurrent streams",
                  "c5": "psd_project phase: psd_sign_lds_kernel<64> (80 blocks of n = 55, one workgroup per block) | psd_sign_wave_kernel<1, 8> (159 blocks of n = 10)"}[args.config]
        kname = {"c1": "lg_sign_cluster_kernel", "c2": "psd_sign_closed_cu_kernel<2" if batched else ("psd_sign_closed_kernel<2" if closed else "psd_sign_wave_kernel<2"),
                 "c3": "lg_gemm_sym_kernel", "c4": "psd_sign_closed_kernel<3" if closed else "psd_sign_wave_kernel<3", "c5": "psd_sign_lds_kernel"}[args.config]
        per_s = psd_ms * 1e-3
        traffic = pmc_traffic(kname, args.config)
        if traffic is not None:
            traffic = traffic[1] * iters_per_launch if (batched and traffic[1]) else traffic[0]
        shard_iters = (world if (args.scaling == "weak" or replicas) else 1) * args.steps
        out = {
            "metric": {"c1": "ADMM iters/sec, PlanarHand_N=1 moment relaxation (+ PSD-proj TFLOP/s in roofline)",
                       "c5": "ADMM iters/sec, pendulum N=80 trajectory SDP (+ PSD-proj TFLOP/s in roofline)",
                       "c2": "ADMM iters/sec, 10k x 32-blk synthetic per GPU (+ PSD-proj TFLOP/s in roofline)",
                       "c3": "ADMM iters/sec, max-cut n=2000 single block (+ PSD-proj TFLOP/s in roofline)",
                       "c4": "ADMM iters/sec, 100k mixed moment-SOS blocks (+ PSD-proj TFLOP/s in roofline)"}[args.config],
            "value": shard_iters / dt,
            "unit": "iters/s (one iteration over a shard; N shards advance together)" if args.scaling == "weak" or replicas
                    else "iters/s of the whole job",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak" if (args.scaling == "weak" or replicas) else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic" if args.config in ("c2", "c3", "c4") else "reference example inputs (rebuilt from the shipped .mat, tests/golden/problems)",
            "config": {"workload": workload,
                       "blocks_total": int(np.asarray(prob.blk).size) * (world if replicas else 1), "vec_len": int(prob.vec_len), "con_num": int(prob.con_num),
                       "sharding": "single GPU" if world == 1 and not force_dist else (
                           "replicas only (one block does not shard)" if replicas else (
                               "blocks by index; all-reduce of [A X | sums | A(S-C)] (2m+2 doubles) before every replicated host y-solve"
                               if args.sharding == "allreduce" else
                               "`value`: blocks by index, constraints owned by the rank whose blocks they touch when the problem is block-diagonal "
                               "(one all-reduce of 4 scalars per iteration, or per launch of several), else the 2m+2 all-reduce (DESIGN.md section 5); "
                               "`allreduce_path` (c2 / c4): the same problem on the general path, 2m+2 doubles all-reduced before every replicated y-solve")),
                       "comm": args.comm if use_comm else None, "init_s": t_init},
            # `achieved` uses the ALGORITHMIC flops of SURVEY 8d (10.67 n^3 per block, what an eigendecomposition-based
            # projection needs) over the measured launch time of the projection; the flops the kernels really issue on the
            # matrix cores for the measured per-block Newton-Schulz step counts are reported beside it.
            "roofline": {"kernel": kernel, "bound": "mfma",
                         "achieved": nominal_flops / per_s / 1e12 if per_s > 0 else 0.0,
                         "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": (nominal_flops / per_s / 1e12 / FP64_PEAK_TFLOPS) if per_s > 0 else 0.0,
                         "traffic": traffic, "avg_launch_ms": psd_ms * iters_per_launch,
                         "launches": int(launches), "iterations_per_launch": iters_per_launch, "ms_per_iteration": psd_ms,
                         # what the matrix cores deliver when EVERY SIMD runs nothing but independent v_mfma_f64_16x16x4_f64
                         # (tools/ubench/mfma_sustained.hip, profiles/r02_mfma_sustained.log): the achievable ceiling
                         "sustained_mfma_peak_measured": SUSTAINED_FP64_MFMA_TFLOPS,
                         "mfma_issued_frac_of_sustained": (issued_flops / per_s / 1e12 / SUSTAINED_FP64_MFMA_TFLOPS) if per_s > 0 else 0.0,
                         "algorithmic_flops_per_launch": nominal_flops * iters_per_launch,
                         "mfma_issued_tflops": issued_flops / per_s / 1e12 if per_s > 0 else 0.0,
                         "mfma_pipe_util": (issued_flops / per_s / 1e12 / FP64_PEAK_TFLOPS) if per_s > 0 else 0.0,
                         "newton_schulz_steps": {"mean": float(steps_blk[sign_blocks].mean()) if sign_blocks.any() else 0.0,
                                                 "max": int(steps_blk.max()) if steps_blk.size else 0,
                                                 "note": "per-block adaptive schedule (csrc/sign_sched.h); round 1 ran a fixed 44"},
                         "hbm_gbs": alg_bytes / per_s / 1e9 if per_s > 0 else 0.0, "algorithmic_bytes_per_launch": alg_bytes * iters_per_launch,
                         "note": "fp64 matrix-core bound (DESIGN.md section 4); traffic = FETCH_SIZE*2 + WRITE_SIZE from the "
                                 "committed rocprofv3 PMC passes (profiles/), for launches of several iterations the profile's bytes "
                                 "per iteration x this run's iterations per launch; algorithmic bytes 32 B per svec element of a closed "
                                 "block (X, C read; S, X written) + its 1.5 KB record, 68 B of a fused block, 16 B elsewhere",
                         "blocks_per_s": blk_local.size / per_s if per_s > 0 else 0.0},
            "final_state": {k: st[k] for k in ("errRp", "errRd", "relgap", "sig")},
        }
        out["steady_state"] = {"value": (world if (args.scaling == "weak" or replicas) else 1) * ss_steps / dt_ss, "steps": ss_steps,
                               "ms_per_step": dt_ss / ss_steps * 1e3,
                               "note": "supplementary: the same solver continued for this many more iterations after the timed region (clock ramp over)"}
        out["pcie_inclusive"] = {"value": (world if (args.scaling == "weak" or replicas) else 1) * args.steps / (t_init + dt + t_readback),
                                 "init_s": t_init, "readback_s": t_readback,
                                 "note": "supplementary: init (host arrays uploaded, A A^T factored) + the timed iterations + X, y, S read back to host arrays"}
        if ck is not None:
            out["with_checkpoint"] = {"value": (world if (args.scaling == "weak" or replicas) else 1) * ck[0] / ck[1], "steps": ck[0], "ms_per_step": ck[1] / ck[0] * 1e3,
                                      "note": "supplementary: continued with stop_tol = 1e-300 (never met): every launch of several iterations starts from a "
                                              "device-side checkpoint of X, S, y, [A X | A (S - C)], as in any solve with a tolerance; `value` (stop_tol = 0) takes none"}
        if ar_path is not None:
            out["allreduce_path"] = ar_path
        out["engine_plan"] = plan
        if args.time_to_tol:
            out["time_to_tol"] = time_to_tol
        if breakdown is not None:
            out["breakdown_ms_per_iter"] = breakdown      # psd_project / aty_xb / post_proj / spmv_A / copies / comm / host / tail_solve
        if breakdown is not None and plan.get("tail_k", 0) > 0 and "tail_solve" in breakdown:
            # supplementary, coupled problems (c1 / c5): the second roofline of their iteration.  The GPU tail of the A A^T solve reads the
            # packed lower triangle of inv(L22) once per solve (tail_solve.hip: 4 K^2 bytes), HBM-bound; the class time beside it also holds
            # the leading sweeps, the tree tops and the L21 products
            k_tail = float(plan["tail_k"])
            solves = 1 if args.mode == "admm" else 2
            out["y_solve"] = {"ms_per_iteration": breakdown["tail_solve"], "solves_per_iteration": solves, "tail_k": int(k_tail),
                              "tail_bytes_per_solve": 4.0 * k_tail * k_tail, "tail_hbm_floor_ms": solves * 4.0 * k_tail * k_tail / 8e12 * 1e3,
                              # N > 1: the tail's rows are split over the ranks (1 / N of the triangle's entries each, tail_solve.h)
                              "tail_bytes_read_per_solve_rank0": solver.tail_info()["bytes_read_per_solve"],
                              "tail_bytes_resident_rank0": solver.tail_info()["bytes_resident"],
                              "plan": {0: "host", 1: "device sweeps + GPU tail", 2: "hybrid (L11 on the host)", 3: "device sweeps + dense tree tops + GPU tail"}
                              .get(int(plan.get("dev_solve", 0)), "?")}
        if world == 1 and not args.no_cpu_baseline:
            threads = usable_cpus()
            out["cpu_baseline"] = cpu_baseline(prob, threads, args.cpu_budget_s)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
