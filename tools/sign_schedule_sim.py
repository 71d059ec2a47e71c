"""Development aid: step counts of the adaptive matrix-sign schedule (psd_sign_lds.h / psd_large.hip) on the spectra that
ADMM actually produces.  The iteration acts on eigenvalues independently, so the schedule is simulated on the
eigenvalues of Xb (exact arithmetic) -- the state machine below is the one the kernels implement.

    python tools/sign_schedule_sim.py c2 300          # synthetic 300 x 32x32, iterations 1..150
    python tools/sign_schedule_sim.py planarhand
"""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))

LIFT_MU = 1.53
CAP = 60


def schedule_steps(lam, lift0=3, lift_more=4, probe=3, trace=False, norm1=None):
    """Return (#steps, max error of sign-weighted eigenvalues relative to ||X||_1-like scale)."""
    lam = np.asarray(lam, float)
    scale = norm1 if norm1 is not None else np.abs(lam).max() * 1.0
    if scale == 0:
        return 0, 0.0
    s = np.abs(lam) / scale
    G = 1.0
    mode, left = "lift", lift0
    steps = 0
    final = False
    while steps < CAP:
        y = s * s
        d2 = np.sum((1 - y) ** 2)
        g = np.sqrt(np.sum((s * (1 - y)) ** 2))
        # exit test (evaluated before this step's update, which is then plain and the last one)
        last = g <= 2.3e-7 and (d2 < 0.5 or g <= 1e-13 * G)
        if mode == "lift":
            if d2 < 0.7:
                mode = "final"
            elif left == 0:
                mode, left = "probe", probe
        if mode == "probe" and left == 0:
            frac = abs(d2 - round(d2))
            if d2 < 0.5 or frac > 1e-3:
                mode = "final"
            else:
                mode, left = "lift", lift_more
        mu = LIFT_MU if (mode == "lift" and not last) else 1.0
        if mode in ("lift", "probe"):
            left -= 1
        s = 1.5 * mu * s - 0.5 * mu ** 3 * s ** 3
        G *= 1.5 * mu
        steps += 1
        if trace:
            print(steps, mode, "d2=%.4g g=%.3g G=%.3g min s=%.3g" % (d2, g, G, s.min()))
        if last:
            break
    err = np.max(np.abs(lam) / scale * (1 - s) / 2)
    return steps, err


def spectra_from_oracle(problem, iters, sample_every=10):
    from oracle import cuadmm_oracle as orc
    out = []

    def eig_fn(bidx, xb):
        x, eigs = orc.psd_project_svec(bidx, xb, return_eigs=True)
        eig_fn.k += 1
        if (eig_fn.k - 1) % sample_every == 0:
            mats = bidx.unpack(xb)
            n1 = [np.abs(M).sum(axis=1).max(axis=1) for M in mats]
            out.append((eig_fn.k, eigs, n1))
        return x
    eig_fn.k = 0
    s = orc.OracleSolver(eig_fn=eig_fn)
    s.init(problem.vec_len, problem.con_num, problem.At_col_ptrs, problem.At_row_ids, problem.At_vals, problem.b_idx,
           problem.b_vals, problem.C_idx, problem.C_vals, problem.blk)
    s.solve(iters, 0.0, 500, 50, 100, 0)
    return out


def report_engine(tag, samples):
    """Step counts of the ENGINE's state machine (csrc/sign_sched.h through the C ABI's host model) on the sampled spectra: per
    block the iterate starts at X / ||X||_F (one-wavefront kernels) -- with the mega-lift of round 5 and without it (mode + 8), in
    the one-wavefront form (statistics of the current iterate) and the lagged form of the batched-GEMM path."""
    import ctypes as C
    import cuadmm_amd
    lib = cuadmm_amd.load()

    def run(spec, mode):
        v = np.ascontiguousarray(np.abs(spec), dtype=np.float64).copy()
        err = C.c_double()
        return lib.cuadmm_sign_sched_simulate(v.ctypes.data_as(C.c_void_p), int(v.size), mode, C.byref(err)), err.value

    for k, eigs, n1 in samples:
        rows = []
        for w in eigs:
            for row in w:
                nf = np.sqrt(np.sum(row * row))
                if nf == 0:
                    continue
                sp = row / nf
                rows.append((len(row), run(sp, 8)[0], run(sp, 0)[0], run(sp, 9)[0], run(sp, 1)[0], max(run(sp, 0)[1], run(sp, 1)[1])))
        r = np.array(rows, float)
        big = r[:, 0] > 16
        print("%s it %4d: %5d blocks  steps (rounds 2-4 -> mega-lift)  wave %.2f -> %.2f   lagged %.2f -> %.2f   | n > 16 (%d blocks): wave %.2f -> %.2f, "
              "lagged %.2f -> %.2f   max err %.1e" % (tag, k, len(r), r[:, 1].mean(), r[:, 2].mean(), r[:, 3].mean(), r[:, 4].mean(), int(big.sum()),
                                                   r[big, 1].mean() if big.any() else 0, r[big, 2].mean() if big.any() else 0,
                                                   r[big, 3].mean() if big.any() else 0, r[big, 4].mean() if big.any() else 0, r[:, 5].max()))


def report(tag, samples, **kw):
    for k, eigs, n1 in samples:
        st, er = [], []
        for w, nn in zip(eigs, n1):
            for row, nrm in zip(w, nn):
                a, b = schedule_steps(row, norm1=nrm, **kw)
                st.append(a); er.append(b)
        st = np.array(st)
        print("%s it %4d: blocks %5d  steps mean %.2f  p50 %d  p90 %d  max %d   max rel err %.2e" %
              (tag, k, st.size, st.mean(), np.percentile(st, 50), np.percentile(st, 90), st.max(), max(er)))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "c2"
    if what == "c2":
        from cuadmm_amd.synthetic import config_c2
        nb = int(sys.argv[2]) if len(sys.argv) > 2 else 200
        p = config_c2(nb)
        samp = spectra_from_oracle(p, 150, 15)
        report("c2", samp)
        report_engine("c2", samp)
    elif what == "c4":
        from cuadmm_amd.synthetic import config_c4
        p = config_c4(600)
        report("c4", spectra_from_oracle(p, 100, 20))
    elif what == "c3":
        from cuadmm_amd.synthetic import config_c3
        p = config_c3(int(sys.argv[2]) if len(sys.argv) > 2 else 400)
        samp = spectra_from_oracle(p, 100, 10)
        report("c3", samp)
        report_engine("c3", samp)
    elif what == "trace":
        rng = np.random.default_rng(0)
        lam = rng.standard_normal(32)
        lam[:3] = 0
        print(schedule_steps(lam, trace=True))
    else:
        from tests.conftest import load_npz_problem as load_golden_problem
        p = load_golden_problem(what)
        iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
        samp = spectra_from_oracle(p, iters, max(1, iters // 6))
        report(what, samp)
        report_engine(what, samp)
