"""GPU parity of the full iteration (SDPSolver::init/solve through the C ABI) against the oracle
trajectories (tests/golden/oracle_traj.json) and the reference's shipped console logs.

Tolerance: per-iteration (errRp, errRd, pobj, dobj, relgap) <= 1e-9 relative (+1e-12 absolute on
quantities at the roundoff floor) over the first iterations, sigma exact; vs the printed log
digits: identical strings where the oracle itself matches the log.
"""
import numpy as np
import pytest

import cuadmm_amd
from oracle import cuadmm_oracle as orc
from tests.conftest import load_npz_problem
from tests.helpers import problem_to_amd

pytestmark = pytest.mark.gpu


def _run(p, iters, sw, stop_tol=0.0, verbose=False):
    s = cuadmm_amd.SDPSolver(verbose=verbose)
    s.init_problem(problem_to_amd(p))
    s.solve(iters, stop_tol, 0, 50, 100, sw, 1.05)
    return s


def _cmp(name, got, ref, rtol=1e-9, atol=1e-12):
    got, ref = np.asarray(got, float), np.asarray(ref, float)
    assert got.shape == ref.shape, name
    err = np.abs(got - ref) / (atol / rtol + np.abs(ref))
    assert np.max(err) <= rtol, "%s: max rel err %.3e at it %d" % (name, np.max(err), int(np.argmax(err)))


@pytest.mark.parametrize("key", ["hinf12/switch=11000", "hinf12/switch=0", "truss5/switch=11000",
                                 "rose13/switch=11000", "ros_2000/switch=0", "ros_2000/switch=11000",
                                 "cnhil10/switch=11000"])
def test_trajectory_vs_oracle_golden(key, oracle_traj, problem_dirs):
    t = oracle_traj[key]
    p = orc.load_problem_txt(problem_dirs[t["problem"]])
    s = _run(p, t["iters"], t["params"]["switch_admm"])
    assert s.info_iter_num == t["iters"]
    st = s.state()
    assert abs(st["bscale"] - float(t["init"]["bscale"])) <= 1e-13 * st["bscale"]
    assert abs(st["Cscale"] - float(t["init"]["Cscale"])) <= 1e-13 * st["Cscale"]
    # rose13 / cnhil10 have a primal residual at the 1e-15 floor in the sGS phase: compare absolutely there
    for nm in ("errRp", "errRd", "pobj", "dobj", "relgap"):
        ref = np.array([float(v) for v in t[nm]])
        _cmp(key + ":" + nm, s.info_arr(nm), ref, rtol=1e-8, atol=1e-11)
    assert np.array_equal(s.info_arr("sig"), np.array([float(v) for v in t["sig"]]))
    assert abs(np.linalg.norm(s.X) - float(t["X_norm"])) <= 1e-8 * (1 + float(t["X_norm"]))
    assert abs(np.linalg.norm(s.S) - float(t["S_norm"])) <= 1e-8 * (1 + float(t["S_norm"]))


def _rows_from_info(s, its):
    rows = {}
    for it in its:
        if it == 0:
            continue
        rows[it] = tuple(s.info_arr(n)[it - 1] for n in ("errRp", "errRd", "pobj", "dobj", "relgap"))
    return rows


@pytest.mark.parametrize("key,iters", [("ros_2000/cuADMM", 300), ("ros_2000/sGS", 300),
                                       ("PushT_N=10_MOMENT/cuADMM", 100), ("PushT_N=10_MOMENT/sGS", 100),
                                       ("rose13/sGS", 200)])
def test_printed_log_digits(key, iters, ref_logs, problem_dirs):
    """The rows the reference printed (examples/benchmarks/**.log) are reproduced digit for digit."""
    lg = ref_logs[key]
    p = orc.load_problem_txt(problem_dirs[lg["problem"]])
    s = _run(p, iters, lg["params"]["switch_admm"], stop_tol=lg["params"]["stop_tol"])
    sig = s.info_arr("sig")
    for row in lg["rows"]:
        it = int(row[0])
        if it == 0 or it > iters:
            continue
        vals = [s.info_arr(n)[it - 1] for n in ("errRp", "errRd", "pobj", "dobj", "relgap")]
        printed = (orc.LOG_ROW_FMT % (it, vals[0], vals[1], vals[2], vals[3], vals[4], 0.0, sig[it - 1])).split("|")
        want = (" %4d | %s %s | %s %s %s |" % (it, row[1], row[2], row[3].rjust(11), row[4].rjust(11), row[5])).split("|")
        got_nums = printed[1].split() + printed[2].split() + [printed[4].strip()]
        want_nums = [row[1], row[2], row[3], row[4], row[5], row[7]]
        for g, w in zip(got_nums, want_nums):
            if float(w) != 0 and abs(float(w)) < 1e-9:
                continue                                                  # roundoff-floor quantities (errRp ~1e-15)
            assert g == w or abs(float(g) - float(w)) <= 1.001e-4 * abs(float(w)), (key, it, got_nums, want_nums)
        del want


def test_warm_restart_if_first_false(problem_dirs):
    """solve(K1) then solve(K2, if_first=false) (solver.cu:385-409: the iterate is re-scaled, residual
    vectors rebuilt, the iteration counter and hence the sigma schedule restart) -- same two calls on the oracle."""
    p = orc.load_problem_txt(problem_dirs["hinf12"])
    o = orc.OracleSolver().init_problem(p)
    o.solve(12, 0.0, 0, 50, 100, 11000, 1.05)
    o.solve(18, 0.0, 0, 50, 100, 11000, 1.05, if_first=False)
    b = cuadmm_amd.SDPSolver(verbose=False)
    b.init_problem(problem_to_amd(p))
    b.solve(12, 0.0, 0, 50, 100, 11000, 1.05)
    b.solve(18, 0.0, 0, 50, 100, 11000, 1.05, if_first=False)
    assert b.info_arr("pobj").size == 30 and b.info_iter_num == 18
    for nm, ref in (("pobj", o.info.pobj), ("dobj", o.info.dobj), ("errRp", o.info.errRp), ("errRd", o.info.errRd)):
        _cmp("restart:" + nm, b.info_arr(nm), np.array(ref), rtol=1e-8, atol=1e-11)
    assert np.array_equal(b.info_arr("sig"), np.array(o.info.sig))
    assert np.max(np.abs(b.X - o.X)) <= 1e-8 * (1 + np.max(np.abs(o.X)))
    assert np.max(np.abs(b.S - o.S)) <= 1e-8 * (1 + np.max(np.abs(o.S)))


def test_init_rejects_bad_input():
    s = cuadmm_amd.SDPSolver(verbose=False)
    with pytest.raises(cuadmm_amd.CuadmmError):
        s.init(15, 30, 7, 1, [0, 1], [0], [1.0], 1, [0], [1.0], 1, [0], [1.0], 1, [3], 1)   # vec_len != 6
    with pytest.raises(cuadmm_amd.CuadmmError):
        cuadmm_amd.SDPSolver(verbose=False).solve(1, 1e-3)                                  # solve before init


def test_converges_like_reference_ros2000(ref_logs, problem_dirs):
    """ADMM-only run to stop_tol=1e-3: same iteration count as examples/benchmarks/ros_2000/cuADMM.log."""
    lg = ref_logs["ros_2000/cuADMM"]
    p = orc.load_problem_txt(problem_dirs["ros_2000"])
    s = _run(p, 10 ** 6, 0, stop_tol=1e-3)
    last_it = int(lg["rows"][-1][0])
    assert abs(s.info_iter_num - last_it) <= max(2, last_it // 200)
    st = s.state()
    assert abs(st["pobj"] - float(lg["final"]["pobj"])) <= 1e-3 * abs(float(lg["final"]["pobj"]))
    assert abs(st["dobj"] - float(lg["final"]["dobj"])) <= 1e-3 * abs(float(lg["final"]["dobj"]))


def test_planarhand_config1_shapes():
    """BASELINE config 1 data (PlanarHand_N=1, 10 block sizes incl. n=120) steps through the engine;
    first printed rows of examples/benchmarks/PlanarHand_N=1_MOMENT/cuADMM.log."""
    p = load_npz_problem("PlanarHand_N=1_MOMENT")
    s = _run(p, 100, 0)
    e50 = [s.info_arr(n)[49] for n in ("errRp", "errRd", "pobj", "dobj", "relgap")]
    want = [1.73e-02, 2.01e-02, 2.9184e-01, 1.8059e+00, 4.89e-01]
    for g, w in zip(e50, want):
        assert abs(g - w) <= 6e-3 * abs(w)


def test_planarhand_converges_at_the_reference_iteration(ref_logs):
    """BASELINE config 1 data end to end (ADMM-only CLI parameters): the engine stops at the iteration the reference
    stopped at (878, examples/benchmarks/PlanarHand_N=1_MOMENT/cuADMM.log) with the same final objective values --
    every projection kernel class, the split A*A^T factor and its GPU tail in one run."""
    lg = ref_logs["PlanarHand_N=1_MOMENT/cuADMM"]
    p = load_npz_problem("PlanarHand_N=1_MOMENT")
    s = _run(p, 20000, 0, stop_tol=1e-3)
    assert s.info_iter_num == int(lg["rows"][-1][0]) == 878
    st = s.state()
    # objective values are O(1e-5 .. 1e-3) on data of size O(1): the reference's own eigensolver tolerance (1e-6) shows
    assert abs(st["pobj"] - float(lg["final"]["pobj"])) <= 2e-7
    assert abs(st["dobj"] - float(lg["final"]["dobj"])) <= 2e-7
    for g, w in zip((st["errRp"], st["errRd"], st["relgap"]), (float(lg["rows"][-1][1]), float(lg["rows"][-1][2]), float(lg["rows"][-1][5]))):
        assert abs(g - w) <= 6e-3 * abs(w)


def test_duo_solver_front(problem_dirs):
    """SDPDuoSolver::init/solve (duo_solver.h:236-276): two block sizes accepted, anything else rejected
    (analyze_blk.cu:39-43); the iteration is the generic one (ros_2000 has sizes {4, 6})."""
    p = orc.load_problem_txt(problem_dirs["ros_2000"])
    a = problem_to_amd(p)
    s = cuadmm_amd.SDPSolver(verbose=False)
    s.duo_init(True, 1, 15, 30, a.vec_len, a.con_num, a.At_csc_col_ptrs, a.At_csc_row_ids, a.At_csc_vals, a.At_nnz,
               a.b_indices, a.b_vals, a.b_nnz, a.C_indices, a.C_vals, a.C_nnz, a.blk_vals, a.mat_num, sig=1.0)
    s.solve(20, 0.0, 0, 50, 100, 11000, 1.05)
    o = orc.OracleSolver().init_problem(p)
    info = o.solve(20, 0.0, 0, 50, 100, 11000, 1.05)
    _cmp("duo:pobj", s.info_arr("pobj"), np.array(info.pobj), rtol=1e-8, atol=1e-11)
    # the reference's host-LAPACK mode is refused loudly (no CPU projection here) unless the caller opts into the GPU kernels
    with pytest.raises(cuadmm_amd.CuadmmError, match="if_gpu_eig_mom"):
        cuadmm_amd.SDPSolver(verbose=False).duo_init(False, 1, 15, 30, a.vec_len, a.con_num, a.At_csc_col_ptrs, a.At_csc_row_ids, a.At_csc_vals,
                                                     a.At_nnz, a.b_indices, a.b_vals, a.b_nnz, a.C_indices, a.C_vals, a.C_nnz, a.blk_vals, a.mat_num, sig=1.0)
    s3 = cuadmm_amd.SDPSolver(verbose=False, options={"duo_cpu_eig_on_gpu": 1})
    s3.duo_init(False, 1, 15, 30, a.vec_len, a.con_num, a.At_csc_col_ptrs, a.At_csc_row_ids, a.At_csc_vals, a.At_nnz,
                a.b_indices, a.b_vals, a.b_nnz, a.C_indices, a.C_vals, a.C_nnz, a.blk_vals, a.mat_num, sig=1.0)
    s3.solve(20, 0.0, 0, 50, 100, 11000, 1.05)
    assert np.array_equal(s3.info_arr("pobj"), s.info_arr("pobj"))
    q = orc.load_problem_txt(problem_dirs["rose13"])            # a single size -> rejected like the reference's assert
    b = problem_to_amd(q)
    with pytest.raises(cuadmm_amd.CuadmmError):
        cuadmm_amd.SDPSolver(verbose=False).duo_init(True, 1, 15, 30, b.vec_len, b.con_num, b.At_csc_col_ptrs, b.At_csc_row_ids,
                                                     b.At_csc_vals, b.At_nnz, b.b_indices, b.b_vals, b.b_nnz, b.C_indices,
                                                     b.C_vals, b.C_nnz, b.blk_vals, b.mat_num)


@pytest.mark.parametrize("name,sw", [("ros_2000", 11000), ("ros_2000", 0), ("pendulum_N=80", 11000), ("biggs", 0)])
def test_duo_solver_n_devices_from_one_process(name, sw, problem_dirs):
    """duo_init(if_gpu_eig_mom = true, device_num_requested = 2) from ONE process (duo_solver.cu:487-577, check_gpus.cu:29-43): the
    handle leads a group of two engines on two host threads, blocks sharded by index, the exchange step an in-process all-reduce
    (csrc/duo_group.hip); option duo_share_device = 1 puts both engines on device 0 (the GPU box has one).  Same trajectory as the
    single engine at 1e-9, X / y / S of the caller's handle are the WHOLE vectors.  ros_2000: block-diagonal (owned constraints,
    four scalars exchanged); pendulum N = 80 (159 x 10 + 80 x 55) and biggs (47 x 13 + 91): coupled (2m+2 doubles, replicated
    solve with the GPU tail of the factor on every rank)."""
    a = problem_to_amd(load_npz_problem(name) if name.startswith("pendulum") else orc.load_problem_txt(problem_dirs[name]))
    iters = 30

    def run(ndev):
        s = cuadmm_amd.SDPSolver(verbose=False, options={"duo_share_device": 1})
        s.duo_init(True, ndev, 15, 30, a.vec_len, a.con_num, a.At_csc_col_ptrs, a.At_csc_row_ids, a.At_csc_vals, a.At_nnz,
                   a.b_indices, a.b_vals, a.b_nnz, a.C_indices, a.C_vals, a.C_nnz, a.blk_vals, a.mat_num, sig=1.0)
        s.solve(iters, 0.0, 0, 50, 100, sw, 1.05)
        return s
    one, two = run(1), run(2)
    assert two.info_iter_num == one.info_iter_num == iters
    assert two.shard() == (0, a.vec_len, 0, a.mat_num) and two.dims() == (a.vec_len, a.con_num, a.mat_num)
    for nm in ("errRp", "errRd", "pobj", "dobj", "relgap"):
        _cmp("duo2:" + nm, two.info_arr(nm), one.info_arr(nm), rtol=1e-9, atol=1e-12)
    assert np.array_equal(two.info_arr("sig"), one.info_arr("sig"))
    for va, vb in ((two.X, one.X), (two.y, one.y), (two.S, one.S)):
        assert va.shape == vb.shape and np.max(np.abs(va - vb)) <= 1e-9 * (1 + np.max(np.abs(vb)))
    if name.startswith("pendulum") or sw == 0 and name == "ros_2000":
        # EIGHT engines from one process (the node the reference's device walk is written for, check_gpus.cu:29-43): eight host threads,
        # eight block ranges, the device-side exchange summing eight staging buffers in rank order, the dense tail split eight ways
        eight = run(8)
        assert eight.group_info()["engines"] == 8 and eight.group_info()["exchange"] == "device"
        for nm in ("errRp", "errRd", "pobj", "dobj", "relgap"):
            _cmp("duo8:" + nm, eight.info_arr(nm), one.info_arr(nm), rtol=1e-9, atol=1e-12)
        assert np.array_equal(eight.info_arr("sig"), one.info_arr("sig"))
        assert np.max(np.abs(eight.X - one.X)) <= 1e-9 * (1 + np.max(np.abs(one.X)))
    # a warm restart through the leader reaches every rank
    two.solve(10, 0.0, 0, 50, 100, sw, 1.05, if_first=False)
    one.solve(10, 0.0, 0, 50, 100, sw, 1.05, if_first=False)
    _cmp("duo2:pobj (continued)", two.info_arr("pobj"), one.info_arr("pobj"), rtol=1e-9, atol=1e-12)
    # more engines than devices without the option: refused, with the way out in the message
    with pytest.raises(cuadmm_amd.CuadmmError, match="duo_share_device"):
        cuadmm_amd.SDPSolver(verbose=False).duo_init(True, 64, 15, 30, a.vec_len, a.con_num, a.At_csc_col_ptrs, a.At_csc_row_ids, a.At_csc_vals,
                                                     a.At_nnz, a.b_indices, a.b_vals, a.b_nnz, a.C_indices, a.C_vals, a.C_nnz, a.blk_vals, a.mat_num, sig=1.0)


def _duo_pendulum(options):
    a = problem_to_amd(load_npz_problem("pendulum_N=80"))
    s = cuadmm_amd.SDPSolver(verbose=False, options=dict({"duo_share_device": 1}, **options))
    s.duo_init(True, 2, 15, 30, a.vec_len, a.con_num, a.At_csc_col_ptrs, a.At_csc_row_ids, a.At_csc_vals, a.At_nnz,
               a.b_indices, a.b_vals, a.b_nnz, a.C_indices, a.C_vals, a.C_nnz, a.blk_vals, a.mat_num, sig=1.0)
    return s


def test_duo_group_exchanges_through_device_memory_by_default():
    """The group's all-reduce: by default each rank's kernel adds the ranks' device staging buffers in rank order (peer reads --
    one shared device here, xGMI peers on a node: the reference's P2P copies, check_gpus.cu:29-43, duo_solver.cu:598-606); option
    duo_exchange = 0 forces the host-staged fallback.  Both sum in rank order: the trajectories are IDENTICAL bit for bit."""
    runs = {}
    for ex in (-1, 0, 1):
        s = _duo_pendulum({"duo_exchange": ex})
        s.solve(25, 0.0, 0, 50, 100, 11000, 1.05)
        gi = s.group_info()
        assert gi["engines"] == 2 and gi["distinct_devices"] == 1 and gi["allreduces"] > 25
        assert gi["exchange"] == ("host" if ex == 0 else "device")
        runs[ex] = [s.info_arr(n).copy() for n in ("errRp", "errRd", "pobj", "dobj", "relgap", "sig")] + [s.X, s.y, s.S]
    for ex in (0, 1):
        for va, vb in zip(runs[ex], runs[-1]):
            assert np.array_equal(va, vb)
    assert cuadmm_amd.SDPSolver(verbose=False).group_info()["engines"] == 1


@pytest.mark.parametrize("exchange", [0, 1])
@pytest.mark.parametrize("inject", [1 * 1000000 + 7, 7, -(1 * 1000000 + 7)])
def test_duo_group_rank_failure_does_not_hang(exchange, inject):
    """A rank of the in-process group fails in the middle of a solve (test hook duo_inject_fail: rank r's k-th collective returns an
    error -- or, negative, throws std::bad_alloc on the rank's host thread): every other rank leaves its barrier with an error, the
    threads are joined, solve() returns a code instead of hanging or calling std::terminate, and the handle can still be destroyed.
    Rank 1 (a child thread) and rank 0 (the caller's thread) both."""
    import time
    s = _duo_pendulum({"duo_exchange": exchange, "duo_inject_fail": inject})
    t0 = time.time()
    with pytest.raises(cuadmm_amd.CuadmmError) as ei:
        s.solve(50, 0.0, 0, 50, 100, 11000, 1.05)
    assert time.time() - t0 < 60.0
    assert "duo group, rank" in str(ei.value)
    if inject < 0:
        assert "bad_alloc" in str(ei.value)
    del s


@pytest.mark.parametrize("exchange", [0, 1])
def test_duo_group_is_reusable_after_a_rank_failed_between_collectives(exchange):
    """A rank fails BETWEEN two collectives (hook + 5e8: in front of its next collective, without counting it) while its peer has
    already entered that collective: the ranks' call counters are one apart when the call returns.  The group is built to be used
    again (duo_group_run clears the abort flag), and the device-side exchange derives its staging slot from that counter -- before
    round 6 the next solve would have summed a STALE staging buffer of the peer (the length check passes: 2m+2 every time).  Now every
    call re-agrees the counters and publishes the collective's index next to its length.  After the failure the iterate is reset on
    every rank (set_XyS) and the solve must reproduce a fresh group's trajectory bit for bit."""
    a = problem_to_amd(load_npz_problem("pendulum_N=80"))
    rng = np.random.default_rng(5)
    X0, S0, y0 = rng.standard_normal(a.vec_len) * 1e-2, rng.standard_normal(a.vec_len) * 1e-2, rng.standard_normal(a.con_num) * 1e-2

    def restart(s):
        s.set_XyS(X0, y0, S0, 1.0)
        s.solve(12, 0.0, 0, 50, 100, 11000, 1.05, if_first=False)
        n = s.info_iter_num
        return [s.info_arr(k)[-n:].copy() for k in ("errRp", "errRd", "pobj", "dobj", "relgap", "sig")] + [s.X, s.y, s.S]

    fresh = _duo_pendulum({"duo_exchange": exchange})
    fresh.solve(9, 0.0, 0, 50, 100, 11000, 1.05)
    want = restart(fresh)
    s = _duo_pendulum({"duo_exchange": exchange})
    s.solve(9, 0.0, 0, 50, 100, 11000, 1.05)
    # rank 1 returns an error in front of its next collective (the solve's first one, batch_agree: no iteration has run on either rank,
    # so the sigma schedule's counters stay equal to the fresh group's) while rank 0 enters and counts it
    s.set_option("duo_inject_fail", 500000000 + 1 * 1000000 + 1)
    with pytest.raises(cuadmm_amd.CuadmmError, match="duo group, rank"):
        s.solve(9, 0.0, 0, 50, 100, 11000, 1.05, if_first=False)
    s.set_option("duo_inject_fail", 0)
    got = restart(s)
    for va, vb in zip(got, want):
        assert np.array_equal(va, vb)


def test_device_side_y_solve_matches_the_host_solve(monkeypatch):
    """Block-diagonal A A^T (every constraint touches one block): the elimination forest of the factor is one small tree per
    block and the y-solve runs on the device, one thread per tree (forest_solve_kernel), with y, A X, A(S-C) and b resident
    in HBM.  Inside a tree it is the serial host algorithm with unfused multiply-subtract, so the whole trajectory must be
    identical to roundoff to the host solve (option host_solve = 1) -- sGS phase, the switch and the ADMM phase included."""
    from cuadmm_amd.synthetic import make_synthetic
    p = make_synthetic([32] * 300 + [6] * 100 + [45] * 40, cons_per_block=4, seed=13)
    prob = cuadmm_amd.Problem(p.vec_len, p.con_num, p.blk, p.At_col_ptrs, p.At_row_ids, p.At_vals, p.b_idx, p.b_vals, p.C_idx, p.C_vals)
    runs = {}
    for mode in ("device", "host"):
        s = cuadmm_amd.SDPSolver(verbose=False, profile=1, options={"host_solve": 1} if mode == "host" else None)
        s.init_problem(prob)
        s.solve(40, 0.0, 0, 10, 10, 15, 1.05)
        s.solve(10, 0.0, 0, 10, 10, 0, 1.05, if_first=False)            # warm restart path
        runs[mode] = (s.X, s.y, s.S, [s.info_arr(n) for n in ("pobj", "dobj", "errRp", "errRd", "relgap", "sig")], s.profile())
    assert runs["device"][4]["host"]["ms"] < 0.2 * runs["host"][4]["host"]["ms"] + 1e-9       # the host solve is gone
    for a, b in zip(runs["device"][:3], runs["host"][:3]):
        assert np.max(np.abs(a - b)) <= 1e-12 * (1 + np.max(np.abs(b)))      # same operation order; the fp64 division differs in the last bit
    for a, b in zip(runs["device"][3], runs["host"][3]):
        assert np.max(np.abs(a - b) / (1e-300 + np.abs(b))) <= 1e-12      # scalars: device / host summation order of the norms


def test_fused_one_launch_sign_kernel_is_bit_identical_across_iterations():
    """psd_lg_fuse (csrc/psd_large.hip: the one-launch sign kernel does its own prologue -- svec -> X0, column sums, S, state -- and stores the svec
    itself, and its barrier counters alternate between two sets because nobody zeroes them before it starts) against the three-launch form, over
    many projections of ONE plan: two mid-size groups that take the fused path (n = 100, 130) beside a 600-block whose one-launch run is NOT fused
    and zeroes counters of its own (a third set: it must not dirty the alternating ones), and small blocks on other kernels.  Same bits."""
    from cuadmm_amd.synthetic import make_synthetic
    q = make_synthetic([100, 600, 130, 100, 20, 7], cons_per_block=6, nnz_per_con=8, seed=77, dense_C=False)
    p = cuadmm_amd.Problem(q.vec_len, q.con_num, q.blk, q.At_col_ptrs, q.At_row_ids, q.At_vals, q.b_idx, q.b_vals, q.C_idx, q.C_vals)
    out = []
    for fuse in (1, 0):
        s = cuadmm_amd.SDPSolver(verbose=False, options={"psd_lg_fuse": fuse})
        s.init_problem(p)
        s.solve(40, 0.0, 0, 50, 100, 0, 1.05)
        out.append((np.array(s.X), np.array(s.y), np.array(s.S), [np.array(s.info_arr(k)) for k in ("errRp", "errRd", "pobj", "dobj")]))
    (X1, y1, S1, i1), (X0, y0, S0, i0) = out
    assert np.array_equal(X1, X0) and np.array_equal(y1, y0) and np.array_equal(S1, S0)
    for a, b in zip(i1, i0):
        assert np.array_equal(a, b)
    assert np.all(np.isfinite(X1)) and i1[0][-1] < i1[0][0]
