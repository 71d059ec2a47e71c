#!/usr/bin/env python3
"""Regenerates tests/golden/* from the read-only reference checkout (run in the dev container only).

    python tests/golden/make_golden.py [/root/reference]

What is produced (all of it DATA -- inputs and expected outputs -- never reference source):

  problems/<name>/{blk,con_num,At,b,C}.txt.gz   example inputs shipped by the reference
                                                (examples/**/TXT/<name>/), gzip'd verbatim
  problems/<name>.npz                           inputs whose At.txt is missing from the checkout
                                                (.MISSING_LARGE_BLOBS) rebuilt from the shipped .mat:
                                                PlanarHand_N=1_MOMENT (MOSEK struct) and
                                                pendulum N=80_licols (SDPT3 struct)
  ref_logs.json                                 the iteration tables printed in the reference's
                                                console logs (examples/benchmarks/**.log, ...),
                                                transcribed as printed strings
  oracle_traj.json                              per-iteration (errRp,errRd,pobj,dobj,relgap,sig)
                                                of oracle/cuadmm_oracle.py in %.17g, first iterations
  io/*                                          the reference unit tests' own data files (test/data)

The reference cannot be compiled here (CUDA + CHOLMOD + MATLAB); its logs
and hard-coded test vectors are the known answers that pin the oracle.
"""
import gzip
import json
import os
import re
import shutil
import sys

import numpy as np
import scipy.io as sio

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import cuadmm_oracle as orc   # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
EX = os.path.join(REF, "examples")

TXT_PROBLEMS = {
    "ros_2000": "plato/TXT/ros_2000",
    "rose13": "plato/TXT/rose13",
    "cnhil10": "plato/TXT/cnhil10",
    "hinf12": "dimacs/data/TXT/hinf12",
    "truss5": "dimacs/data/TXT/truss5",
    "PushT_N=10_MOMENT": "SPOT/data/TXT/PushT_N=10_MOMENT",
}

# (log file, problem, mode parameters found by reproduction -- SURVEY.md F7)
LOGS = {
    "ros_2000/cuADMM": ("benchmarks/ros_2000/cuADMM.log", "ros_2000", dict(switch_admm=0)),
    "ros_2000/sGS": ("benchmarks/ros_2000/sGS-cuADMM.log", "ros_2000", dict(switch_admm=11000)),
    "PushT_N=10_MOMENT/cuADMM": ("benchmarks/PushT_N=10_MOMENT/cuADMM.log", "PushT_N=10_MOMENT", dict(switch_admm=0)),
    "PushT_N=10_MOMENT/sGS": ("benchmarks/PushT_N=10_MOMENT/sGS-cuADMM.log", "PushT_N=10_MOMENT", dict(switch_admm=11000)),
    "PlanarHand_N=1_MOMENT/cuADMM": ("benchmarks/PlanarHand_N=1_MOMENT/cuADMM.log", "PlanarHand_N=1_MOMENT", dict(switch_admm=0)),
    "PlanarHand_N=1_MOMENT/sGS": ("benchmarks/PlanarHand_N=1_MOMENT/sGS-cuADMM.log", "PlanarHand_N=1_MOMENT", dict(switch_admm=11000)),
    "rose13/sGS": ("plato/logs/rose13.log", "rose13", dict(switch_admm=11000)),
    "cnhil10/sGS": ("plato/logs/cnhil10.log", "cnhil10", dict(switch_admm=11000)),
    "pendulum_N=80/sGS": ("pendulum/N=80_licols.log", "pendulum_N=80", dict(switch_admm=11000)),
    # single large blocks (n = 1024 / 861 / 800): the large-block projection path against the reference's own runs
    "1dc.1024/sGS": ("plato/logs/1dc.1024.log", "1dc.1024", dict(switch_admm=11000)),
    "bqp-r1-40-1/sGS": ("plato/logs/bqp-r1-40-1.log", "bqp-r1-40-1", dict(switch_admm=11000)),
    "swissroll/sGS": ("plato/logs/swissroll.log", "swissroll", dict(switch_admm=11000)),
    # inputs that exist only as MOSEK / svec .mat files (round 2: long-run parity on problems never run before)
    "PushT_N=30_MOMENT/sGS": ("benchmarks/PushT_N=30_MOMENT/sGS-cuADMM.log", "PushT_N=30_MOMENT", dict(switch_admm=11000)),
    "chs_5000/cuADMM": ("benchmarks/chs_5000/cuADMM.log", "chs_5000", dict(switch_admm=0)),
    "chs_5000/sGS": ("benchmarks/chs_5000/sGS-cuADMM.log", "chs_5000", dict(switch_admm=11000)),
}

# problems shipped as .mat with At / C / b already in svec form (examples/plato/MATLAB), blk from the TXT directory
PLATO_MAT = {
    "1dc.1024": ("plato/MATLAB/1dc.1024.mat", "plato/TXT/1dc.1024"),
    "bqp-r1-40-1": ("plato/MATLAB/bqp-r1-40-1.mat", "plato/TXT/bqp-r1-40-1"),
    "swissroll": ("plato/MATLAB/swissroll.mat", "plato/TXT/swissroll"),
    "chs_5000": ("plato/MATLAB/chs_5000.mat", "plato/TXT/chs5000"),
    # SeDuMi (At, b, c, K) with 14 blocks of 56 / 126 / 252; no cuADMM log is shipped, only MOSEK's and ADMM+'s: the fixture carries
    # MOSEK's optimum (benchmarks/taha1a/MOSEK.log).  The TXT directory's b.txt has the OPPOSITE sign of the .mat's b (its At.txt is
    # missing: it was written from a copy with A and b both negated -- the same feasible set); the .mat is what is converted.
    "taha1a": ("plato/MATLAB/taha1a.mat", "plato/TXT/taha1a"),
}
MOSEK_OPT = {"taha1a": ("benchmarks/taha1a/MOSEK.log", -1.0000000103e+00, -1.0000000154e+00)}   # primal, dual objective of the log's summary
COMMON = dict(sig=1.0, stop_tol=1e-3, sig_update_threshold=0, sig_update_stage_1=50,
              sig_update_stage_2=100, sigscale=1.05)

ROW = re.compile(r"^\s*(\d+) \| (\S+) (\S+) \|\s+(\S+)\s+(\S+) (\S+) \|\s*(\S+) \| (\S+) \|\s*$")


def gz_copy(src, dst):
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(src, "rb") as f, gzip.GzipFile(dst, "wb", mtime=0) as g:
        shutil.copyfileobj(f, g)


def transcribe_log(path):
    rows, header, final = [], {}, {}
    with open(path) as f:
        lines = f.read().splitlines()
    for ln in lines:
        m = ROW.match(ln)
        if m:
            rows.append(list(m.groups()))
            continue
        for key, pat in (("vec_len", r"vector length: (\d+)"), ("con_num", r"number of constraints: (\d+)"),
                         ("mat_num", r"number of blocks: (\d+)"), ("At_nnz", r"non-zeros in At: (\d+)"),
                         ("b_nnz", r"non-zeros in b: (\d+)"), ("C_nnz", r"non-zeros in C: (\d+)")):
            mm = re.search(pat, ln)
            if mm:
                header[key] = int(mm.group(1))
        mm = re.search(r"norm of C = (\S+), norm of b = (\S+)", ln)
        if mm:
            header["norm_C"], header["norm_b"] = mm.group(1), mm.group(2)
        for key, pat in (("pinf", r"primal infeasibility = (\S+)"), ("dinf", r"dual   infeasibility = (\S+)"),
                         ("relgap", r"relative gap         = (\S+)"), ("pobj", r"primal objective = \s*(\S+)"),
                         ("dobj", r"dual   objective = \s*(\S+)")):
            mm = re.search(pat, ln)
            if mm:
                final[key] = mm.group(1)
        if "Solver ended" in ln:
            final["msg"] = ln.strip()
    census = [ln for ln in lines if re.search(r"matrices of size", ln)]
    return dict(header=header, rows=rows, final=final, census=census)


def mosek_to_problem(matpath):
    """MOSEK struct -> svec TXT triplets (examples/mosek_to_txt.m, utils/convert_mosek2sedumi.m:35-50,
    utils/read_sedumi.m:128-154, sedumi_to_txt.m:60-73 restated; validated on PushT_N=10)."""
    d = sio.loadmat(matpath, squeeze_me=True, struct_as_record=False)
    prob = d["prob"]
    bardim = np.atleast_1d(prob.bardim).astype(np.int64)
    off = orc.svec_block_offsets(bardim)
    bara = prob.bara
    subi = np.atleast_1d(bara.subi).astype(np.int64) - 1
    subj = np.atleast_1d(bara.subj).astype(np.int64) - 1
    subk = np.atleast_1d(bara.subk).astype(np.int64) - 1
    subl = np.atleast_1d(bara.subl).astype(np.int64) - 1
    val = np.atleast_1d(bara.val).astype(np.float64)
    rows = off[subj] + subk * (subk + 1) // 2 + subl
    vals = np.where(subk != subl, val * np.sqrt(2.0), val)
    m = int(np.atleast_1d(prob.blc).size)
    barc = prob.barc
    cj = np.atleast_1d(barc.subj).astype(np.int64) - 1
    ck = np.atleast_1d(barc.subk).astype(np.int64) - 1
    cl = np.atleast_1d(barc.subl).astype(np.int64) - 1
    cv = np.atleast_1d(barc.val).astype(np.float64)
    crow = off[cj] + ck * (ck + 1) // 2 + cl
    cval = np.where(ck != cl, cv * np.sqrt(2.0), cv)
    b = np.atleast_1d(prob.blc).astype(np.float64)
    return dict(blk=bardim.astype(np.int32), con_num=m, At_row=rows.astype(np.int32), At_col=subi.astype(np.int32),
                At_val=vals, C_idx=crow.astype(np.int32), C_val=cval,
                b_idx=np.nonzero(b)[0].astype(np.int32), b_val=b[np.nonzero(b)[0]])


def main():
    out_prob = os.path.join(HERE, "problems")
    os.makedirs(out_prob, exist_ok=True)
    for name, rel in TXT_PROBLEMS.items():
        for fn in ("blk.txt", "con_num.txt", "At.txt", "b.txt", "C.txt"):
            gz_copy(os.path.join(EX, rel, fn), os.path.join(out_prob, name, fn + ".gz"))

    # --- PlanarHand_N=1 from the MOSEK struct; cross-check b/C/blk against the shipped TXT
    ph = mosek_to_problem(os.path.join(EX, "SPOT/data/MOSEK/PlanarHand_N=1_MOMENT.mat"))
    txt = os.path.join(EX, "SPOT/data/TXT/PlanarHand_N=1_MOMENT/")
    assert [n for _, n in orc.read_blk(txt + "blk.txt")] == ph["blk"].tolist()
    ci, cv = orc.read_sparse_vector(txt + "C.txt")
    o = np.argsort(ph["C_idx"])
    assert np.array_equal(ph["C_idx"][o], ci) and np.allclose(ph["C_val"][o], cv, rtol=0, atol=1e-15)
    bi, bv = orc.read_sparse_vector(txt + "b.txt")
    assert np.array_equal(ph["b_idx"], bi) and np.allclose(ph["b_val"], bv, rtol=0, atol=1e-15)
    ph["C_idx"], ph["C_val"] = ci, cv          # keep the shipped text values
    ph["b_idx"], ph["b_val"] = bi, bv
    assert ph["At_val"].size == 156635         # benchmarks/PlanarHand_N=1_MOMENT/cuADMM.log:5
    np.savez_compressed(os.path.join(out_prob, "PlanarHand_N=1_MOMENT.npz"), **ph)

    # --- PushT_N=30 (the A*A^T factorisation stress test: 369 s before iteration 0 in the reference's log), same transform
    p30 = mosek_to_problem(os.path.join(EX, "SPOT/data/MOSEK/PushT_N=30_MOMENT.mat"))
    txt = os.path.join(EX, "SPOT/data/TXT/PushT_N=30_MOMENT/")
    assert [n for _, n in orc.read_blk(txt + "blk.txt")] == p30["blk"].tolist()
    ci, cv = orc.read_sparse_vector(txt + "C.txt")
    o = np.argsort(p30["C_idx"])
    assert np.array_equal(p30["C_idx"][o], ci) and np.allclose(p30["C_val"][o], cv, rtol=0, atol=1e-15)
    bi, bv = orc.read_sparse_vector(txt + "b.txt")
    assert np.array_equal(p30["b_idx"], bi) and np.allclose(p30["b_val"], bv, rtol=0, atol=1e-15)
    p30["C_idx"], p30["C_val"], p30["b_idx"], p30["b_val"] = ci, cv, bi, bv
    np.savez_compressed(os.path.join(out_prob, "PushT_N=30_MOMENT.npz"), **p30)

    # the same transform reproduces the shipped PushT_N=10 At.txt (validation of the transform)
    pt = mosek_to_problem(os.path.join(EX, "SPOT/data/MOSEK/PushT_N=10_MOMENT.mat"))
    r, c, v = orc.read_coo(os.path.join(EX, "SPOT/data/TXT/PushT_N=10_MOMENT/At.txt"))
    k1 = np.lexsort((pt["At_row"], pt["At_col"])); k2 = np.lexsort((r, c))
    assert np.array_equal(pt["At_row"][k1], r[k2]) and np.array_equal(pt["At_col"][k1], c[k2])
    assert np.max(np.abs(pt["At_val"][k1] - v[k2])) < 1e-15

    # --- pendulum N=80 from the SDPT3 struct (At already in svec form)
    d = sio.loadmat(os.path.join(EX, "pendulum/MATLAB/N=80_licols.mat"), squeeze_me=True, struct_as_record=False)
    sdp = d["SDP"].sdpt3
    import scipy.sparse as sp
    At = sdp.At                 # cell array: one svec-form sparse matrix per block, in blk order
    At = sp.vstack(list(np.atleast_1d(At))).tocoo() if not hasattr(At, "tocoo") else At.tocoo()
    ptxt = os.path.join(EX, "pendulum/TXT/N=80_licols/")
    blk = np.array([n for _, n in orc.read_blk(ptxt + "blk.txt")], dtype=np.int32)
    bi, bv = orc.read_sparse_vector(ptxt + "b.txt")
    ci, cv = orc.read_sparse_vector(ptxt + "C.txt")
    con_num = int(open(ptxt + "con_num.txt").read().split()[0])
    assert At.shape == (int(orc.svec_block_offsets(blk)[-1]), con_num) and At.nnz == 278569
    np.savez_compressed(os.path.join(out_prob, "pendulum_N=80.npz"), blk=blk, con_num=con_num,
                        At_row=At.row.astype(np.int32), At_col=At.col.astype(np.int32), At_val=At.data,
                        C_idx=ci, C_val=cv, b_idx=bi, b_val=bv)

    # --- single-large-block problems from examples/plato/MATLAB (their TXT directories lack At.txt / C.txt)
    for name, (matrel, txtrel) in PLATO_MAT.items():
        d = sio.loadmat(os.path.join(EX, matrel))
        blk = np.array([n for _, n in orc.read_blk(os.path.join(EX, txtrel, "blk.txt"))], dtype=np.int32)
        if "At" not in d or "K" in d:      # SeDuMi (A | At, b, c, K): through the converter (cuadmm_amd/convert.py = examples/sedumi_to_txt.m)
            from cuadmm_amd import convert
            import scipy.sparse as sp
            q = convert.problem_from_sedumi_mat(os.path.join(EX, matrel))
            assert list(q.blk_vals) == blk.tolist()
            At = sp.csc_matrix((q.At_csc_vals, q.At_csc_row_ids, q.At_csc_col_ptrs), shape=(q.vec_len, q.con_num)).tocoo()
            bi_txt, bv_txt = orc.read_sparse_vector(os.path.join(EX, txtrel, "b.txt"))
            ci_txt, cv_txt = orc.read_sparse_vector(os.path.join(EX, txtrel, "C.txt"))
            b_sign = -1.0 if name == "taha1a" else 1.0          # see PLATO_MAT
            assert np.array_equal(q.b_indices, bi_txt) and np.allclose(q.b_vals, b_sign * bv_txt, rtol=0, atol=1e-12)
            assert np.array_equal(q.C_indices, ci_txt) and np.allclose(q.C_vals, cv_txt, rtol=0, atol=1e-12)
            extra = {}
            if name in MOSEK_OPT:
                logrel, pw, dw = MOSEK_OPT[name]
                txt = open(os.path.join(EX, logrel)).read()
                assert ("Primal.  obj: %.10e" % pw) in txt and ("Dual.    obj: %.10e" % dw) in txt
                extra = dict(mosek_pobj=pw, mosek_dobj=dw)
            np.savez_compressed(os.path.join(out_prob, name + ".npz"), blk=blk, con_num=int(q.con_num),
                                At_row=At.row.astype(np.int32), At_col=At.col.astype(np.int32), At_val=At.data,
                                C_idx=ci_txt, C_val=cv_txt, b_idx=bi_txt, b_val=b_sign * bv_txt, **extra)
            continue
        At, Cm = d["At"].tocoo(), d["C"].tocoo()
        bm = np.asarray(d["b"].todense() if hasattr(d["b"], "todense") else d["b"]).ravel()
        assert At.shape[0] == int(orc.svec_block_offsets(blk)[-1]) and At.shape[1] == bm.size
        bi_txt, bv_txt = orc.read_sparse_vector(os.path.join(EX, txtrel, "b.txt"))
        bi = np.nonzero(bm)[0].astype(np.int32)
        assert np.array_equal(bi, bi_txt) and np.allclose(bm[bi], bv_txt, rtol=0, atol=1e-12)
        o = np.argsort(Cm.row)
        np.savez_compressed(os.path.join(out_prob, name + ".npz"), blk=blk, con_num=int(At.shape[1]),
                            At_row=At.row.astype(np.int32), At_col=At.col.astype(np.int32), At_val=At.data,
                            C_idx=Cm.row[o].astype(np.int32), C_val=Cm.data[o], b_idx=bi, b_val=bm[bi])

    # --- source-format inputs of the converters (cuadmm_amd/convert.py): the expected outputs are the shipped TXT
    # directories (problems/hinf12, truss5, PushT_N=10_MOMENT above; biggs below)
    fmt = os.path.join(HERE, "formats")
    os.makedirs(fmt, exist_ok=True)
    shutil.copyfile(os.path.join(EX, "dimacs/data/MATLAB/hinf12.mat"), os.path.join(fmt, "hinf12_sedumi.mat"))
    shutil.copyfile(os.path.join(EX, "dimacs/data/MATLAB/truss5.mat"), os.path.join(fmt, "truss5_sedumi.mat"))
    shutil.copyfile(os.path.join(EX, "SPOT/data/MOSEK/PushT_N=10_MOMENT.mat"), os.path.join(fmt, "PushT_N=10_MOMENT_mosek.mat"))
    shutil.copyfile(os.path.join(EX, "plato/MATLAB/rose13.mat"), os.path.join(fmt, "rose13_svec.mat"))
    gz_copy(os.path.join(EX, "plato/MATLAB/biggs.dat-s"), os.path.join(fmt, "biggs.dat-s.gz"))
    for fn in ("blk.txt", "con_num.txt", "At.txt", "b.txt", "C.txt"):
        gz_copy(os.path.join(EX, "plato/TXT/biggs", fn), os.path.join(out_prob, "biggs", fn + ".gz"))

    # --- transcribed logs
    logs = {}
    for key, (rel, prob, mode) in LOGS.items():
        t = transcribe_log(os.path.join(EX, rel))
        if len(t["rows"]) > 16:              # keep the fixture small: head of the table + its last rows
            t["rows"] = t["rows"][:14] + t["rows"][-2:]
        t["problem"] = prob
        t["params"] = dict(COMMON, **mode)
        t["source"] = "examples/" + rel
        logs[key] = t
    with open(os.path.join(HERE, "ref_logs.json"), "w") as f:
        json.dump(logs, f, indent=1)

    # --- reference unit-test data files
    os.makedirs(os.path.join(HERE, "io"), exist_ok=True)
    for fn in os.listdir(os.path.join(REF, "test/data")):
        shutil.copyfile(os.path.join(REF, "test/data", fn), os.path.join(HERE, "io", fn))

    # --- oracle trajectories (full precision) on the small problems
    traj = {}
    for name, iters, sw in (("hinf12", 40, 11000), ("hinf12", 40, 0), ("truss5", 40, 11000),
                            ("rose13", 30, 11000), ("ros_2000", 30, 0), ("ros_2000", 30, 11000),
                            ("cnhil10", 30, 11000)):
        p = orc.load_problem_txt(os.path.join(EX, TXT_PROBLEMS[name]))
        s = orc.OracleSolver().init_problem(p)
        info = s.solve(iters, 0.0, 0, 50, 100, sw, 1.05)
        key = f"{name}/switch={sw}"
        traj[key] = dict(problem=name, iters=iters,
                         params=dict(COMMON, switch_admm=sw, stop_tol=0.0),
                         init=dict(norm_borg=repr(s.norm_borg), norm_Corg=repr(s.norm_Corg),
                                   bscale=repr(s.bscale), Cscale=repr(s.Cscale)),
                         errRp=[repr(x) for x in info.errRp], errRd=[repr(x) for x in info.errRd],
                         pobj=[repr(x) for x in info.pobj], dobj=[repr(x) for x in info.dobj],
                         relgap=[repr(x) for x in info.relgap], sig=[repr(x) for x in info.sig],
                         X_norm=repr(float(np.linalg.norm(s.X))), y_norm=repr(float(np.linalg.norm(s.y))),
                         S_norm=repr(float(np.linalg.norm(s.S))))
    with open(os.path.join(HERE, "oracle_traj.json"), "w") as f:
        json.dump(traj, f, indent=1)
    print("golden fixtures written under", HERE)


if __name__ == "__main__":
    main()
