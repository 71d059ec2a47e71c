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


@pytest.mark.parametrize("args,port", [
    (["--blocks-per-gpu", "600", "--steps", "12", "--warmup", "3"], 29651),                              # C2, owned constraints, batched launches
    (["--blocks-per-gpu", "600", "--steps", "6", "--warmup", "2", "--sharding", "allreduce"], 29652),   # C2 over the general path
    (["--config", "c5", "--steps", "8", "--warmup", "2", "--time-to-tol", "0"], 29653),                  # pendulum N = 80: replicated device-side solve
    (["--config", "c1", "--steps", "6", "--warmup", "2", "--time-to-tol", "0"], 29654),                  # PlanarHand_N=1
    (["--config", "c4", "--blocks-per-gpu", "1200", "--steps", "6", "--warmup", "2"], 29655),            # mixed sizes, strong scaling
])
def test_bench_two_ranks_on_one_gpu(args, port):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CUADMM_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline"] + args,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                 # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["steps"] == int(args[args.index("--steps") + 1])
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


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` AS TYPED (no torch.distributed.run): the parent starts the two rank processes itself
    (bench.launch_ranks; the one-process launch of the reference's src/duo_solver.cu:487-577) and relays rank 0's line."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CUADMM_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline", "--blocks-per-gpu", "600",
                        "--steps", "12", "--warmup", "3"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["steps"] == 12
    assert d["allreduce_path"]["value"] > 0
