"""`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), on ONE GPU: the ranks share
device 0 and the all-reduce goes through gloo (CUADMM_BENCH_BACKEND=gloo) -- everything else is the code path of the multi-GPU
runs: argument handling, sharding (owned constraints for the block-diagonal C2, the 2m+2 all-reduce for it when forced and for
the coupled moment relaxations c1 / c5), barrier + max-over-ranks timing, the one JSON line of rank 0."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_line(d, n, args):
    assert d["n_gpus"] == n and d["value"] > 0 and d["steps"] == int(args[args.index("--steps") + 1])
    assert d["roofline"]["frac"] > 0 and d["config"]["comm"] == "torch"
    assert d["scaling"] == ("weak" if "--config" not in args else "strong")
    block_diagonal = "--config" not in args or "c4" in args
    if block_diagonal and "--sharding" not in args:
        # `value` took the owned-constraints shortcut; the collective the north star names is in the same line
        ar = d["allreduce_path"]
        assert ar["value"] > 0 and ar["allreduce_ms_per_iter"] > 0 and ar["allreduce_doubles"] == 2 * d["config"]["con_num"] + 2
        assert "owned" in d["config"]["sharding"] and "allreduce_path" in d["config"]["sharding"]
        assert "--config" in args or d["engine_plan"]["closed_blocks"] == 1.0     # C2: every block closed on every rank
        assert d["with_checkpoint"]["value"] > 0
    else:
        assert "allreduce_path" not in d
    if "c1" in args or "c5" in args:
        # the replicated y-solve: its dense tail split N ways (rank 0's share of the 4 K^2 bytes per solve; tail_shard_bound)
        ys = d["y_solve"]
        K = (ys["tail_k"] + 63) // 64 * 64
        assert abs(ys["tail_bytes_read_per_solve_rank0"] - 4.0 * K * K / n) <= 64.0 * K + 512, ys
        assert d["breakdown_ms_per_iter"]["allreduce"] > 0


@pytest.mark.parametrize("args,port", [
    (["--blocks-per-gpu", "600", "--steps", "6", "--warmup", "2", "--sharding", "allreduce"], 29652),   # C2 over the general path
])
def test_bench_two_ranks_on_one_gpu(args, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CUADMM_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline"] + args,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                 # ONE line, from rank 0
    _check_line(json.loads(lines[0]), 2, args)


@pytest.mark.parametrize("n,args", [
    (8, ["--blocks-per-gpu", "600", "--steps", "12", "--warmup", "3"]),                                   # C2 weak: owned constraints (batched launches) + allreduce_path
    (8, ["--config", "c4", "--blocks-per-gpu", "4800", "--scaling", "strong", "--steps", "6", "--warmup", "2"]),   # mixed sizes, ONE problem over 8 ranks
    (8, ["--config", "c1", "--steps", "6", "--warmup", "2", "--time-to-tol", "0"]),                      # PlanarHand_N=1: replicated solve, tail split 8 ways
    (8, ["--config", "c5", "--steps", "8", "--warmup", "2", "--time-to-tol", "0"]),                      # pendulum N = 80
    (4, ["--blocks-per-gpu", "600", "--steps", "12", "--warmup", "3"]),
])
def test_bench_gpus_n_as_typed_on_one_gpu(n, args):
    """`python bench.py --gpus 8` / `--gpus 4` AS TYPED (the parent starts the rank processes itself, bench.launch_ranks -- the
    one-process launch of the reference's src/duo_solver.cu:487-577; a launcher's form is test_bench_two_ranks_on_one_gpu): all ranks
    share device 0, the collective goes through gloo.  Everything else is the code path of the 8-GPU runs the driver makes: sharding
    of 8 x 600 blocks / of one problem, owned constraints and the 2m+2 all-reduce, batch_agree, the replicated y-solve with its tail
    split 8 ways, barrier + max-over-ranks timing, ONE JSON line."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CUADMM_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--no-cpu-baseline"] + args,
                       env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    _check_line(json.loads(lines[0]), n, args)
