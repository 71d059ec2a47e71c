"""Phase ticks of the one-wavefront sign kernels for a bulk of equal blocks (CUADMM_PSD_DEBUG=1: prologue / iteration / epilogue per block).
python tools/probe_class_ticks.py [n count]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
os.environ["CUADMM_PSD_DEBUG"] = "1"
from helpers import psd_project_gpu  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 45
count = int(sys.argv[2]) if len(sys.argv) > 2 else 16667
rng = np.random.default_rng(n)
blk = np.full(count, n, np.int32)
x = rng.standard_normal(count * n * (n + 1) // 2)
for _ in range(3):
    t0 = time.perf_counter()
    psd_project_gpu(x, blk)
    sys.stderr.write("  call %.2f ms (with copies)\n" % ((time.perf_counter() - t0) * 1e3))
