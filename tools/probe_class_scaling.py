"""Kernel time of one size class against its block count (does a launch of many rounds cost more than rounds x one round?).
Run under rocprofv3 --kernel-trace --stats: python tools/probe_class_scaling.py n count [count ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from helpers import psd_project_gpu  # noqa: E402

n = int(sys.argv[1])
for count in [int(a) for a in sys.argv[2:]]:
    rng = np.random.default_rng(n)
    blk = np.full(count, n, np.int32)
    x = rng.standard_normal(count * n * (n + 1) // 2)
    for _ in range(3):
        psd_project_gpu(x, blk)
