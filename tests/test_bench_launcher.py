"""bench.py's own rank launcher (`python bench.py --gpus N` as typed) on a box WITHOUT a GPU: the ranks rendezvous over gloo,
find no device and exit with an error -- the launcher must come back with a non-zero code and no JSON line instead of hanging."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_returns_the_ranks_failure():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: tests/test_gpu_bench_ranks.py runs the launcher for real")
    env = dict(os.environ, CUADMM_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--no-cpu-baseline", "--blocks-per-gpu", "50",
                        "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "rank exit codes" in r.stderr
