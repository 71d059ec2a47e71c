"""Host-side mirror of the reference's operator interface for the hot path.

``SDPSolver`` has the reference's method pair and member names (include/cuadmm/solver.h:30-248):
``init(...)`` / ``solve(...)`` with the same argument lists and defaults, then ``X``, ``y``, ``S``
(unscaled after solve), ``info_iter_num``, ``info_*_arr`` and ``total_time``.  ``Problem.from_txt``
mirrors Problem::from_txt (src/problem.cu:11-83).  Everything is executed by libcuadmm_amd.so.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check

_INFO = {"pobj": 0, "dobj": 1, "errRp": 2, "errRd": 3, "relgap": 4, "sig": 5, "bscale": 6, "Cscale": 7}


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Problem:
    """TXT problem directory (blk.txt, con_num.txt, At.txt, b.txt, C.txt), src/problem.cu:11-83."""

    def __init__(self, vec_len, con_num, blk_vals, At_csc_col_ptrs, At_csc_row_ids, At_csc_vals,
                 b_indices, b_vals, C_indices, C_vals):
        self.vec_len, self.con_num = int(vec_len), int(con_num)
        self.blk_vals = _i32(blk_vals)
        self.mat_num = int(self.blk_vals.size)
        self.At_csc_col_ptrs, self.At_csc_row_ids = _i32(At_csc_col_ptrs), _i32(At_csc_row_ids)
        self.At_csc_vals = _f64(At_csc_vals)
        self.At_nnz = int(self.At_csc_vals.size)
        self.b_indices, self.b_vals = _i32(b_indices), _f64(b_vals)
        self.C_indices, self.C_vals = _i32(C_indices), _f64(C_vals)
        self.b_nnz, self.C_nnz = int(self.b_vals.size), int(self.C_vals.size)

    @classmethod
    def from_txt(cls, prefix):
        lib = _lib.load()
        h = C.c_void_p()
        check(lib.cuadmm_problem_from_txt(prefix.encode(), C.byref(h)))
        try:
            v = _lib.ProblemView()
            check(lib.cuadmm_problem_view_get(h, C.byref(v)))
            arr = lambda ptr, n, dt: np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True) if n > 0 else np.zeros(0, dt)
            return cls(v.vec_len, v.con_num, arr(v.blk_vals, v.mat_num, np.int32),
                       arr(v.At_csc_col_ptrs, v.con_num + 1, np.int32), arr(v.At_csc_row_ids, v.At_nnz, np.int32),
                       arr(v.At_csc_vals, v.At_nnz, np.float64), arr(v.b_indices, v.b_nnz, np.int32),
                       arr(v.b_vals, v.b_nnz, np.float64), arr(v.C_indices, v.C_nnz, np.int32),
                       arr(v.C_vals, v.C_nnz, np.float64))
        finally:
            lib.cuadmm_problem_free(h)

    @classmethod
    def from_coo(cls, blk, con_num, At_row, At_col, At_val, b_idx, b_val, C_idx, C_val):
        """COO triplets of At (svec_row, constraint_col, value) -> sorted CSC (COO_to_CSC, io.cu:187-243)."""
        lib = _lib.load()
        blk = _i32(blk)
        b64 = blk.astype(np.int64)
        vec_len = int(np.sum(np.where(b64 >= 0, b64 * (b64 + 1) // 2, -b64)))     # negative size: unconstrained block
        rows, cols, vals = _i32(At_row).copy(), _i32(At_col).copy(), _f64(At_val).copy()
        cp = np.zeros(con_num + 1, np.int32)
        check(lib.cuadmm_coo_to_csc(_p(cp), _p(cols), _p(rows), _p(vals), int(vals.size), int(con_num)))
        return cls(vec_len, con_num, blk, cp, rows, vals, b_idx, b_val, C_idx, C_val)


class SDPSolver:
    """Mirror of class SDPSolver (include/cuadmm/solver.h:30-248)."""

    def __init__(self, device=0, verbose=True, rank=0, world=1, profile=False, force_comm=False, psd_steps=False,
                 eig_rank=0, eig_rank_begin_iter=0, eig_rank_maxfeas=0.0, options=None):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        check(self._lib.cuadmm_create(C.byref(self._h)))
        for k, v in (("device", device), ("verbose", int(bool(verbose))), ("rank", rank), ("world", world),
                     ("profile", int(profile)), ("force_comm", int(bool(force_comm))), ("psd_steps", int(bool(psd_steps))),
                     ("eig_rank", int(eig_rank)), ("eig_rank_begin_iter", int(eig_rank_begin_iter)), ("eig_rank_maxfeas", float(eig_rank_maxfeas))):
            check(self._lib.cuadmm_set_option(self._h, k.encode(), float(v)))
        for k, v in (options or {}).items():           # engine options by name (include/cuadmm_amd.h: cuadmm_set_option)
            self.set_option(k, v)
        self._cb = None
        self.vec_len = self.con_num = 0

    def __del__(self):
        try:
            if self._h:
                self._lib.cuadmm_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_option(self, key, value):
        check(self._lib.cuadmm_set_option(self._h, key.encode(), float(value)))

    def counters(self):
        o = np.zeros(8)
        check(self._lib.cuadmm_get_counters(self._h, _p(o)))
        keys = ["batch_launches", "batch_iters", "batch_rollbacks", "host_pool_threads", "fused", "closed_blocks", "dev_solve", "tail_k"]
        return dict(zip(keys, o.tolist()))

    def group_info(self):
        """The in-process group this handle leads after duo_init(device_num_requested = N) from one process."""
        o = np.zeros(4)
        check(self._lib.cuadmm_get_group_info(self._h, _p(o)))
        return {"engines": int(o[0]), "exchange": "device" if o[1] else "host", "distinct_devices": int(o[2]), "allreduces": int(o[3])}

    def tail_info(self):
        """The dense GPU tail of the A A^T factor on this rank (cuadmm_get_tail_info)."""
        o = np.zeros(6)
        check(self._lib.cuadmm_get_tail_info(self._h, _p(o)))
        return {"tail_k": int(o[0]), "bytes_read_per_solve": float(o[1]), "rows": int(o[2]), "bytes_resident": float(o[3]),
                "inverse_residual": float(o[4]), "refined": bool(o[5])}

    def set_allreduce(self, fn):
        """fn(dev_ptr:int, count:int, hip_stream:int) -> None : in-place sum over ranks on that stream."""
        def tramp(_user, buf, count, stream):
            try:
                fn(int(buf or 0), int(count), int(stream or 0))
                return 0
            except Exception as e:                     # surfaces as CUADMM_ERR_COMM
                import traceback
                traceback.print_exc()
                return 1
        self._cb = _lib.ALLREDUCE_FN(tramp)
        check(self._lib.cuadmm_set_allreduce(self._h, self._cb, None))

    # -- SDPSolver::init (solver.h:208-223) -------------------------------------------------
    def init(self, eig_stream_num_per_gpu, cpu_eig_thread_num, vec_len, con_num,
             At_csc_col_ptrs, At_csc_row_ids, At_csc_vals, At_nnz,
             b_indices, b_vals, b_nnz, C_indices, C_vals, C_nnz, blk_vals, mat_num,
             X=None, y=None, S=None, sig=1.0):
        a = [_i32(At_csc_col_ptrs), _i32(At_csc_row_ids), _f64(At_csc_vals), _i32(b_indices), _f64(b_vals),
             _i32(C_indices), _f64(C_vals), _i32(blk_vals)]
        Xa = None if X is None else _f64(X)
        ya = None if y is None else _f64(y)
        Sa = None if S is None else _f64(S)
        check(self._lib.cuadmm_init(self._h, int(eig_stream_num_per_gpu), int(cpu_eig_thread_num), int(vec_len),
                                    int(con_num), _p(a[0]), _p(a[1]), _p(a[2]), int(At_nnz), _p(a[3]), _p(a[4]),
                                    int(b_nnz), _p(a[5]), _p(a[6]), int(C_nnz), _p(a[7]), int(mat_num),
                                    _p(Xa), _p(ya), _p(Sa), float(sig)))
        self.vec_len, self.con_num, self.mat_num = int(vec_len), int(con_num), int(mat_num)
        return self

    def init_problem(self, p, X=None, y=None, S=None, sig=1.0, eig_stream_num_per_gpu=15, cpu_eig_thread_num=30):
        return self.init(eig_stream_num_per_gpu, cpu_eig_thread_num, p.vec_len, p.con_num, p.At_csc_col_ptrs,
                         p.At_csc_row_ids, p.At_csc_vals, p.At_nnz, p.b_indices, p.b_vals, p.b_nnz,
                         p.C_indices, p.C_vals, p.C_nnz, p.blk_vals, p.mat_num, X, y, S, sig)

    # -- SDPSolver::solve (solver.h:236-244) ------------------------------------------------
    def solve(self, max_iter, stop_tol, sig_update_threshold=500, sig_update_stage_1=50, sig_update_stage_2=100,
              switch_admm=int(1.1e4), sigscale=1.05, if_first=True):
        check(self._lib.cuadmm_solve(self._h, int(max_iter), float(stop_tol), int(sig_update_threshold),
                                     int(sig_update_stage_1), int(sig_update_stage_2), int(switch_admm),
                                     float(sigscale), int(bool(if_first))))
        return self

    # -- SDPDuoSolver::init / ::solve (duo_solver.h:236-276): two block sizes only ---------------------
    def duo_init(self, if_gpu_eig_mom, device_num_requested, eig_stream_num_per_gpu, cpu_eig_thread_num, vec_len, con_num,
                 At_csc_col_ptrs, At_csc_row_ids, At_csc_vals, At_nnz, b_indices, b_vals, b_nnz,
                 C_indices, C_vals, C_nnz, blk_vals, mat_num, X=None, y=None, S=None, sig=2e2):
        a = [_i32(At_csc_col_ptrs), _i32(At_csc_row_ids), _f64(At_csc_vals), _i32(b_indices), _f64(b_vals),
             _i32(C_indices), _f64(C_vals), _i32(blk_vals)]
        Xa = None if X is None else _f64(X)
        ya = None if y is None else _f64(y)
        Sa = None if S is None else _f64(S)
        check(self._lib.cuadmm_duo_init(self._h, int(bool(if_gpu_eig_mom)), int(device_num_requested),
                                        int(eig_stream_num_per_gpu), int(cpu_eig_thread_num), int(vec_len), int(con_num),
                                        _p(a[0]), _p(a[1]), _p(a[2]), int(At_nnz), _p(a[3]), _p(a[4]), int(b_nnz),
                                        _p(a[5]), _p(a[6]), int(C_nnz), _p(a[7]), int(mat_num), _p(Xa), _p(ya), _p(Sa), float(sig)))
        self.vec_len, self.con_num, self.mat_num = int(vec_len), int(con_num), int(mat_num)
        return self

    # -- results (public members of the reference class) -----------------------------------------
    def dims(self):
        """(vec_len, con_num, mat_num) in the caller's numbering (cuadmm_get_dims)."""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        check(self._lib.cuadmm_get_dims(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def shard(self):
        b, e, kb, ke = C.c_int64(), C.c_int64(), C.c_int(), C.c_int()
        check(self._lib.cuadmm_get_shard(self._h, C.byref(b), C.byref(e), C.byref(kb), C.byref(ke)))
        return b.value, e.value, kb.value, ke.value

    def _vec(self, fn, n):
        out = np.empty(n, np.float64)
        check(fn(self._h, _p(out)))
        return out

    @property
    def X(self):
        b, e, _, _ = self.shard()
        return self._vec(self._lib.cuadmm_get_X, e - b)

    @property
    def S(self):
        b, e, _, _ = self.shard()
        return self._vec(self._lib.cuadmm_get_S, e - b)

    @property
    def y(self):
        return self._vec(self._lib.cuadmm_get_y, self.con_num)

    def set_XyS(self, X=None, y=None, S=None, sig=0.0):
        Xa = None if X is None else _f64(X)
        ya = None if y is None else _f64(y)
        Sa = None if S is None else _f64(S)
        check(self._lib.cuadmm_set_XyS(self._h, _p(Xa), _p(ya), _p(Sa), float(sig)))

    @property
    def info_iter_num(self):
        return int(self._lib.cuadmm_get_info_iter_num(self._h))

    def info_arr(self, name):
        cap = 1 << 22
        n_total = 0
        out = np.empty(4096, np.float64)
        n = self._lib.cuadmm_get_info_array(self._h, _INFO[name], _p(out), out.size)
        if n == out.size:
            out = np.empty(cap, np.float64)
            n = self._lib.cuadmm_get_info_array(self._h, _INFO[name], _p(out), out.size)
        del n_total
        return out[:check(n)].copy()

    info_pobj_arr = property(lambda self: self.info_arr("pobj"))
    info_dobj_arr = property(lambda self: self.info_arr("dobj"))
    info_errRp_arr = property(lambda self: self.info_arr("errRp"))
    info_errRd_arr = property(lambda self: self.info_arr("errRd"))
    info_relgap_arr = property(lambda self: self.info_arr("relgap"))
    info_sig_arr = property(lambda self: self.info_arr("sig"))
    info_bscale_arr = property(lambda self: self.info_arr("bscale"))
    info_Cscale_arr = property(lambda self: self.info_arr("Cscale"))

    @property
    def total_time(self):
        return float(self._lib.cuadmm_get_total_time(self._h))

    def state(self):
        o = np.zeros(12)
        check(self._lib.cuadmm_get_state(self._h, _p(o)))
        keys = ["errRp", "errRd", "pobj", "dobj", "relgap", "sig", "bscale", "Cscale", "norm_borg", "norm_Corg",
                "best_KKT", "eig_not_converged"]
        return dict(zip(keys, o.tolist()))

    def profile(self):
        o = np.zeros(24)
        check(self._lib.cuadmm_get_profile(self._h, _p(o)))
        names = ["aty_xb", "psd_project", "post_proj", "spmv_A", "copies", "host", "allreduce", "tail_solve"]
        return {n: dict(launches=o[3 * i], ms=o[3 * i + 1], bytes_per_launch=o[3 * i + 2]) for i, n in enumerate(names)}

    def psd_steps(self):
        """Newton-Schulz steps of the last projection per local block (needs psd_steps=True at construction)."""
        b, e, kb, ke = self.shard()
        out = np.zeros(max(ke - kb, 1), np.int32)
        n = self._lib.cuadmm_get_psd_steps(self._h, _p(out), out.size)
        check(min(n, 0))
        return out[:n]

    def reset_profile(self):
        check(self._lib.cuadmm_reset_profile(self._h))
