"""ctypes binding of libcuadmm_amd.so (the C ABI declared in include/cuadmm_amd.h).

The shared library is the product; this module only declares prototypes.  If the library is
missing it is built in-tree with hipcc (cuadmm_amd.build); if that fails the import fails -- there
is no Python/CPU fallback for the device path.
"""
import ctypes as C
import os

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcuadmm_amd.so")

c_int_p = C.POINTER(C.c_int)
c_double_p = C.POINTER(C.c_double)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class ProblemView(C.Structure):
    _fields_ = [("vec_len", C.c_int), ("con_num", C.c_int), ("mat_num", C.c_int),
                ("At_nnz", C.c_int), ("b_nnz", C.c_int), ("C_nnz", C.c_int),
                ("At_csc_col_ptrs", c_int_p), ("At_csc_row_ids", c_int_p), ("At_csc_vals", c_double_p),
                ("b_indices", c_int_p), ("b_vals", c_double_p), ("C_indices", c_int_p), ("C_vals", c_double_p),
                ("blk_vals", c_int_p)]


# name -> (restype, argtypes); every symbol declared in include/cuadmm_amd.h
PROTOTYPES = {
    "cuadmm_last_error": (C.c_char_p, []),
    "cuadmm_version": (C.c_char_p, []),
    "cuadmm_device_count": (C.c_int, []),
    "cuadmm_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "cuadmm_destroy": (None, [C.c_void_p]),
    "cuadmm_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    "cuadmm_set_allreduce": (C.c_int, [C.c_void_p, ALLREDUCE_FN, C.c_void_p]),
    "cuadmm_rccl_unique_id": (C.c_int, [C.c_char_p]),
    "cuadmm_use_rccl": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int]),
    "cuadmm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                               C.c_void_p, C.c_void_p, C.c_int,
                               C.c_void_p, C.c_void_p, C.c_int,
                               C.c_void_p, C.c_int,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]),
    "cuadmm_duo_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]),
    "cuadmm_duo_solve": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]),
    "cuadmm_solve": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int]),
    "cuadmm_get_dims": (C.c_int, [C.c_void_p, c_int_p, c_int_p, c_int_p]),
    "cuadmm_get_X": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cuadmm_get_y": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cuadmm_get_S": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cuadmm_set_XyS": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double]),
    "cuadmm_get_device_ptrs": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "cuadmm_get_shard": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), c_int_p, c_int_p]),
    "cuadmm_aat_factor_arrays": (C.c_int, [C.c_void_p] + [C.POINTER(C.c_void_p)] * 4),
    "cuadmm_aat_forest": (C.c_int, [C.c_void_p, c_int_p, c_int_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "cuadmm_mex_call": (C.c_int, [C.c_int, C.c_int, C.c_double,
                                   C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_size_t, C.c_void_p,
                                   C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                   C.c_double, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "cuadmm_mex_result_dims": (C.c_int, [C.c_void_p, c_int_p, c_int_p, c_int_p, C.POINTER(C.c_double)]),
    "cuadmm_mex_result_XyS": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cuadmm_mex_result_info": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "cuadmm_mex_result_free": (None, [C.c_void_p]),
    "cuadmm_get_psd_steps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "cuadmm_sign_sched_simulate_hint": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_double_p, c_int_p]),
    "cuadmm_sign_sched_simulate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_double_p]),
    "cuadmm_op_psd_project_ex": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cuadmm_op_psd_project_steps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "cuadmm_get_info_iter_num": (C.c_int, [C.c_void_p]),
    "cuadmm_get_info_array": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "cuadmm_get_total_time": (C.c_double, [C.c_void_p]),
    "cuadmm_get_state": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cuadmm_get_profile": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cuadmm_reset_profile": (C.c_int, [C.c_void_p]),
    "cuadmm_get_counters": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cuadmm_get_group_info": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cuadmm_get_tail_info": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cuadmm_problem_from_txt": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "cuadmm_problem_view_get": (C.c_int, [C.c_void_p, C.POINTER(ProblemView)]),
    "cuadmm_problem_free": (None, [C.c_void_p]),
    "cuadmm_coo_to_csc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "cuadmm_read_blk": (C.c_int, [C.c_char_p, C.c_void_p, C.c_void_p, C.c_int]),
    "cuadmm_write_dense_txt": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int64]),
    "cuadmm_is_large_mat": (C.c_int, [C.c_int, C.c_int]),
    "cuadmm_analyze_blk": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "cuadmm_get_maps": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cuadmm_get_maps_duo": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cuadmm_inverse_permutation": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "cuadmm_partition_blocks": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "cuadmm_aat_create": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.POINTER(C.c_void_p)]),
    "cuadmm_aat_perm": (c_int_p, [C.c_void_p]),
    "cuadmm_aat_factor_nnz": (C.c_int64, [C.c_void_p]),
    "cuadmm_aat_factor_colptr": (C.POINTER(C.c_int64), [C.c_void_p]),
    "cuadmm_aat_solve_permuted": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cuadmm_aat_tail_plan": (C.c_int, [C.c_void_p, C.c_int]),
    "cuadmm_aat_create_split": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.POINTER(C.c_void_p)]),
    "cuadmm_aat_tail_k": (C.c_int, [C.c_void_p]),
    "cuadmm_aat_tail_tops": (C.c_int, [C.c_void_p]),
    "cuadmm_aat_plan_allow_tops": (None, [C.c_int]),
    "cuadmm_aat_tail_schur": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(C.c_int64)), C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_double))]),
    "cuadmm_aat_tail_schur_release": (None, [C.c_void_p]),
    "cuadmm_aat_tail_dense": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "cuadmm_aat_solve_leading_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "cuadmm_aat_solve_leading_backward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "cuadmm_aat_solve_leading_forward11": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "cuadmm_aat_solve_leading_backward11": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "cuadmm_aat_free": (None, [C.c_void_p]),
    "cuadmm_op_vector_to_matrices": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_void_p]),
    "cuadmm_op_matrices_to_vector": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_void_p]),
    "cuadmm_op_gemm_sym": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cuadmm_op_tail_factor_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "cuadmm_op_tail_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "cuadmm_op_tail_solve_sharded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cuadmm_tail_shard_bounds": (C.c_int, [C.c_int, C.c_int, C.c_void_p]),
    "cuadmm_op_tail_solve_drill": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cuadmm_op_batch_eig": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "cuadmm_op_max_zero": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "cuadmm_op_mul_diag_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "cuadmm_op_mul_trans_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "cuadmm_psd_plan_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "cuadmm_psd_plan_project": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cuadmm_psd_plan_destroy": (None, [C.c_void_p]),
    "cuadmm_op_psd_project": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "cuadmm_op_permute": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "cuadmm_op_get_normA": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "cuadmm_op_spmv_csr": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]),
    "cuadmm_op_axpby2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int64, C.c_void_p]),
    "cuadmm_op_axpby3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int64, C.c_void_p]),
    "cuadmm_op_norm2": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_double), C.c_void_p]),
    "cuadmm_dev_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "cuadmm_dev_free": (C.c_int, [C.c_void_p]),
    "cuadmm_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "cuadmm_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "cuadmm_dev_sync": (C.c_int, []),
}

_lib = None


def load(rebuild=False):
    """Returns the loaded library, building it first if needed."""
    global _lib
    if _lib is not None and not rebuild:
        return _lib
    if rebuild or not os.path.exists(LIB_PATH):
        # several ranks of one job may get here together (torch.distributed.run on a fresh checkout): one builds under an
        # exclusive file lock, the others wait and then find the library
        import fcntl
        os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
        with open(LIB_PATH + ".lock", "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            try:
                if rebuild or not os.path.exists(LIB_PATH):
                    _build.build(force=rebuild)
            finally:
                fcntl.flock(lk, fcntl.LOCK_UN)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class CuadmmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("cuadmm_amd error %d: %s" % (code, msg))
        self.code = code


def check(rc):
    if rc < 0:
        raise CuadmmError(rc, load().cuadmm_last_error().decode(errors="replace"))
    return rc
