#!/usr/bin/env python3
"""The reference's LARGEST shipped moment relaxations as fixtures (run in the dev container; reads /root/reference once):

    python tests/golden/make_large_moment.py

examples/SPOT/data/MOSEK/{PushBox_N=30_MOMENT, PushBox_N=50_MOMENT, PlanarHand_N=10_MOMENT}.mat (MOSEK `prob` structs; published
times examples/benchmarks/benchmarks.csv:2-5,46-49) through cuadmm_amd/convert.py (the job of examples/mosek_to_txt.m) into
tests/golden/problems/<name>.npz -- data only: block sizes, the COO of A^T, b, C.  The GPU box has no /root/reference.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from cuadmm_amd import convert   # noqa: E402

SRC = "/root/reference/examples/SPOT/data/MOSEK"
NAMES = ["PushBox_N=30_MOMENT", "PushBox_N=50_MOMENT", "PlanarHand_N=10_MOMENT"]


def main():
    for name in NAMES:
        p = convert.load_any(os.path.join(SRC, name + ".mat"))
        cp = np.asarray(p.At_csc_col_ptrs)
        col = np.repeat(np.arange(p.con_num, dtype=np.int32), np.diff(cp))
        out = os.path.join(HERE, "problems", name + ".npz")
        np.savez_compressed(out, blk=np.asarray(p.blk_vals, np.int32), con_num=int(p.con_num),
                            At_row=np.asarray(p.At_csc_row_ids, np.int32), At_col=col, At_val=np.asarray(p.At_csc_vals, np.float64),
                            C_idx=np.asarray(p.C_indices, np.int32), C_val=np.asarray(p.C_vals, np.float64),
                            b_idx=np.asarray(p.b_indices, np.int32), b_val=np.asarray(p.b_vals, np.float64))
        print(name, "L", p.vec_len, "m", p.con_num, "nnz", p.At_nnz, "->", os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
