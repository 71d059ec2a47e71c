"""Merged one-launch groups against the per-group launches on a given block list (bit identity), repeated.
python tools/dbg/merge_check.py 252 56 56 56 126"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from helpers import psd_project_gpu  # noqa: E402

blk = np.array([int(a) for a in sys.argv[1:]], np.int32)
rng = np.random.default_rng(1)
L = int(np.sum(blk.astype(np.int64) * (blk + 1) // 2))
bad = 0
for rep in range(40):
    x = rng.standard_normal(L)
    os.environ["CUADMM_PSD_LG_MERGE"] = "1"
    a = psd_project_gpu(x, blk)
    os.environ["CUADMM_PSD_LG_MERGE"] = "0"
    b = psd_project_gpu(x, blk)
    if not np.array_equal(a, b):
        bad += 1
        print("rep", rep, "max diff", float(np.max(np.abs(a - b))))
print("mismatches:", bad, "of 40")
