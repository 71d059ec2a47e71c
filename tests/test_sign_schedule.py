"""CPU tests of the per-block adaptive matrix-sign schedule (cuadmm_amd/csrc/sign_sched.h).

The Newton-Schulz iteration acts on the eigenvalues of a block independently, so the state machine the kernels run can
be exercised on spectra alone through the host model exported by the C ABI (no device).  What is pinned here:
termination inside the cap, the resolution contract (every eigenvalue with |lambda| >= 1e-13 ||X||_1 ends with its sign
resolved to <= 1e-13 relative error of the projection; smaller ones contribute at most their own size), and that typical
ADMM spectra take a quarter of the fixed 44-step schedule of round 1.
"""
import ctypes as C

import numpy as np
import pytest

import cuadmm_amd


def _run(spec, lagged=False):
    lib = cuadmm_amd.load()
    s = np.ascontiguousarray(np.abs(spec), dtype=np.float64).copy()
    err = C.c_double(0.0)
    steps = lib.cuadmm_sign_sched_simulate(s.ctypes.data_as(C.c_void_p), int(s.size), int(lagged), C.byref(err))
    return steps, err.value, s


def _spectra(rng):
    out = {}
    n = 32
    w = rng.standard_normal(n)
    out["generic"] = w / (2.5 * np.abs(w).max())
    g = np.logspace(0, -10, n) * np.where(np.arange(n) % 2, 1, -1)
    out["graded"] = g / 1.3
    r = w.copy(); r[:20] = 0.0
    out["rank_deficient_exact"] = r / (2.0 * np.abs(r).max())
    r2 = w.copy(); r2[:20] = 1e-17 * rng.standard_normal(20)
    out["rank_deficient_roundoff"] = r2 / (2.0 * np.abs(r2).max())
    t = w.copy(); t[0] = 1e-13; t[1] = -1e-13; t[2] = 3e-12
    out["pm1e-13"] = t / (1.5 * np.abs(t).max())
    out["identity"] = np.ones(n) / 1.0
    out["zero"] = np.zeros(n)
    out["single"] = np.array([0.7])
    m = np.concatenate([[0.8, 0.3], 10.0 ** rng.uniform(-16, -11, 8)])   # moment-matrix like (PlanarHand blocks)
    out["moment_like"] = m
    out["clustered_small"] = np.concatenate([[1.0], np.full(31, 1e-6)]) / 1.2
    return out


@pytest.mark.parametrize("lagged", [False, True])
def test_schedule_terminates_and_meets_the_resolution_contract(lagged):
    rng = np.random.default_rng(7)
    for name, spec in _spectra(rng).items():
        steps, err, s = _run(spec, lagged)
        assert 1 <= steps <= 64, (name, steps)
        # projection error relative to ||X||_1: resolved eigenvalues to roundoff, unresolved ones <= their own size
        assert err <= 2.5e-13, (name, steps, err)
        big = np.abs(spec) >= 1e-11
        assert np.all(np.abs(1.0 - s[big]) <= 1e-12), (name, steps, s[big])


def test_typical_spectra_take_a_fraction_of_the_fixed_schedule():
    rng = np.random.default_rng(11)
    tot = 0
    for _ in range(300):
        w = rng.standard_normal(32)
        w /= np.abs(w).sum() * 0.35 + np.abs(w).max()      # 1-norm style over-estimate of the spectral radius
        steps, err, _ = _run(w)
        assert err <= 1e-13
        tot += steps
    assert tot / 300 <= 16.0          # fixed schedule of round 1: 44


def test_exactly_rank_deficient_blocks_stop_early():
    rng = np.random.default_rng(3)
    w = rng.standard_normal(32)
    w[:24] = 0.0
    steps, err, _ = _run(w / (3 * np.abs(w).max()))
    assert steps <= 14 and err <= 1e-13


def test_random_spectra_fuzz():
    rng = np.random.default_rng(5)
    worst = 0
    for trial in range(2000):
        n = int(rng.integers(1, 65))
        kind = trial % 4
        if kind == 0:
            w = rng.standard_normal(n)
        elif kind == 1:
            w = 10.0 ** rng.uniform(-18, 0, n) * rng.choice([-1, 1], n)
        elif kind == 2:
            w = rng.standard_normal(n); w[rng.random(n) < 0.5] = 0.0
        else:
            w = np.concatenate([rng.uniform(0.2, 1, n // 2 + 1), 10.0 ** rng.uniform(-15, -3, n)])[:n]
        scale = np.abs(w).max() * rng.uniform(1.0, 6.0)
        if scale == 0:
            scale = 1.0
        for lag in (False, True):
            steps, err, _ = _run(w / scale, lag)
            assert steps <= 64
            assert err <= 2.5e-13, (trial, lag, steps, err)
            worst = max(worst, steps)
    assert worst <= 64


def test_schedule_warm_start_hint_costs_steps_only():
    """lift0 = the number of lift steps the previous projection needed (any value is safe): the resolution contract holds for
    every hint, the right hint saves steps, and the hint a run leaves reproduces its own step count."""
    lib = cuadmm_amd.load()
    rng = np.random.default_rng(21)
    saved = 0
    for trial in range(300):
        n = int(rng.integers(2, 65))
        w = rng.standard_normal(n) if trial % 2 else 10.0 ** rng.uniform(-15, 0, n)
        w = np.abs(w) / (np.abs(w).max() * rng.uniform(1.0, 4.0))
        base, err0, _ = _run(w)
        hint_next = None
        for lift0 in (1, 2, 5, 9, 20, 40, 64):
            s = np.ascontiguousarray(w, dtype=np.float64).copy()
            err, lifts = C.c_double(), C.c_int()
            steps = lib.cuadmm_sign_sched_simulate_hint(s.ctypes.data_as(C.c_void_p), n, 0, lift0, C.byref(err), C.byref(lifts))
            assert 1 <= steps <= 64 and err.value <= 2.5e-13, (trial, lift0, steps, err.value)
        # self-consistency: run with the hint the default run leaves
        s = np.ascontiguousarray(w, dtype=np.float64).copy()
        err, lifts = C.c_double(), C.c_int()
        lib.cuadmm_sign_sched_simulate_hint(s.ctypes.data_as(C.c_void_p), n, 0, 0, C.byref(err), C.byref(lifts))
        s = np.ascontiguousarray(w, dtype=np.float64).copy()
        steps2 = lib.cuadmm_sign_sched_simulate_hint(s.ctypes.data_as(C.c_void_p), n, 0, lifts.value, C.byref(err), C.byref(lifts))
        assert err.value <= 2.5e-13
        saved += base - steps2
    assert saved > 0                      # on average the warm start is a gain


def test_skipping_the_statistics_changes_no_decision():
    """The one-wavefront kernels skip the three wave reductions on the steps whose scale is fixed in advance
    (SignSched::needs_stats: lift phases, bursts, the two probe steps).  Mode 0 of the host model calls the state machine
    the same way; mode 3 passes the statistics on every step: identical step counts and identical iterates, bit for bit."""
    rng = np.random.default_rng(2024)
    specs = list(_spectra(rng).values())
    for _ in range(300):
        n = int(rng.integers(2, 65))
        kind = rng.integers(0, 4)
        if kind == 0:
            w = rng.standard_normal(n)
        elif kind == 1:
            w = 10.0 ** rng.uniform(-16, 0, n) * rng.choice([-1.0, 1.0], n)
        elif kind == 2:
            w = rng.standard_normal(n); w[: n // 2] = 0.0
        else:
            w = rng.standard_normal(n) * 1e-9; w[:2] = rng.uniform(0.3, 1.0, 2)
        specs.append(w / max(np.sqrt((w * w).sum()), 1e-300))
    for w in specs:
        st0, e0, s0 = _run(w, 0)
        st3, e3, s3 = _run(w, 3)
        assert st0 == st3 and e0 == e3 and np.array_equal(s0, s3)


def test_mega_lift_jumps_a_gap_and_leaves_continuous_spectra_alone():
    """Round 5 (sign_sched.h, MEGA-LIFT): on a spectrum with a GAP -- a few eigenvalues of order one, a cluster at 1e-6 ... 1e-11 -- one
    step with the coefficients (1 + c, -c) carries the whole cluster up (c capped at 600 in total: the roundoff it amplifies); mode + 8
    is the schedule of rounds 2-4.  Fewer steps on gapped spectra, never more on Gaussian ones (mostly the same: no gap, no mega-lift --
    which is why the closed-block kernels may run the plain machine), the resolution contract either way, in both forms of the machine."""
    lib = cuadmm_amd.load()
    rng = np.random.default_rng(17)

    def run(spec, mode):
        s = np.ascontiguousarray(np.abs(spec), dtype=np.float64).copy()
        err = C.c_double()
        return lib.cuadmm_sign_sched_simulate(s.ctypes.data_as(C.c_void_p), int(s.size), mode, C.byref(err)), err.value

    saved = {0: 0, 1: 0}
    same = 0
    for trial in range(200):
        n = int(rng.integers(12, 65))
        top = rng.uniform(0.3, 1.0, max(2, n // 5))
        gap = 10.0 ** rng.uniform(-11, -5)
        low = gap * rng.uniform(0.05, 1.0, n - top.size)
        spec = np.concatenate([top, low]) / rng.uniform(1.0, 3.0)
        for lag in (0, 1):
            new, e_new = run(spec, lag)
            old, e_old = run(spec, lag + 8)
            assert e_new <= 2.5e-13 and e_old <= 2.5e-13
            assert new <= old + 3, (trial, lag, new, old)         # (+3: waits for a tighter bound that did not pay: steps, never accuracy)
            saved[lag] += old - new
        w = rng.standard_normal(n)
        w /= np.abs(w).max() * rng.uniform(1.0, 3.0)
        for lag in (0, 1):
            a, b = run(w, lag)[0], run(w, lag + 8)[0]
            assert a <= b                                             # a Gaussian spectrum has a gap now and then (its smallest eigenvalue alone)
            same += a == b
    assert saved[0] >= 200 * 3 and saved[1] >= 200 * 3                # >= 3 steps per gapped block on average
    assert same >= 0.8 * 400                                          # ... but mostly not: same schedule as rounds 2-4


# ---- the CLEAN mega-lift (mode + 16: what the batched-GEMM path's groups padded to <= 512 run; sign_sched.h) -------------------------

def _gap_spectrum(rng, n, r, lo, hi, zeros=0):
    lam = np.zeros(n)
    lam[:r] = rng.uniform(0.2, 1.0, r)
    m = n - r - zeros
    lam[r:r + m] = 10.0 ** rng.uniform(lo, hi, m)
    return lam / 1.1


def test_clean_mega_lift_meets_the_contract_on_every_spectrum_family():
    rng = np.random.default_rng(7)
    for name, spec in _spectra(rng).items():
        steps, err, s = _run(spec, 1 + 16)
        assert 1 <= steps <= 64, (name, steps)
        assert err <= 2.5e-13, (name, steps, err)
        big = np.abs(spec) >= 1e-11
        assert np.all(np.abs(1.0 - s[big]) <= 1e-12), (name, steps, s[big])


def test_clean_mega_lift_jumps_a_gap_the_capped_one_climbs():
    """A gap of 1e-9 ... 1e-11 (a relaxation's numerically low-rank iterate): the capped mega-lift takes 600 of the ~1e9 in one step
    and lifts the rest 2.3-fold per step; the clean one takes it whole for two step slots."""
    rng = np.random.default_rng(17)
    tot = {0: 0, 1: 0, 2: 0}
    for _ in range(40):
        spec = _gap_spectrum(rng, 120, 8, -11, -9)
        none, e0, _ = _run(spec, 1 + 8)
        capped, e1, _ = _run(spec, 1)
        clean, e2, _ = _run(spec, 1 + 16)
        assert max(e0, e1, e2) <= 2.5e-13
        assert clean <= capped and clean <= none - 8, (none, capped, clean)
        tot[0] += none; tot[1] += capped; tot[2] += clean
    assert tot[2] <= 0.8 * tot[1], tot


def test_clean_mega_lift_leaves_gapless_spectra_alone():
    rng = np.random.default_rng(19)
    for _ in range(100):
        w = rng.standard_normal(96)
        w /= np.abs(w).sum() * 0.35 + np.abs(w).max()
        a, _, sa = _run(w, 1)
        b, _, sb = _run(w, 1 + 16)
        assert a == b and np.array_equal(sa, sb)      # the same schedule to the bit: C3's single block and the synthetic classes see no change


def test_clean_mega_lift_fuzz():
    rng = np.random.default_rng(23)
    for t in range(400):
        n = int(rng.integers(65, 130))
        r = int(rng.integers(0, 12))
        lo = float(rng.uniform(-15, -3))
        spec = _gap_spectrum(rng, n, r, lo, lo + float(rng.uniform(0.0, 3.0)), zeros=int(rng.integers(0, 20)))
        spec *= rng.choice([-1.0, 1.0], n)
        steps, err, _ = _run(spec, 1 + 16)
        assert 1 <= steps <= 64 and err <= 2.5e-13, (t, steps, err)
