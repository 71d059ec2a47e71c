"""Loader of oracle/cpu_eig_baseline.c -- the reference's eig_cpu path (per-block LAPACK dsyevd on T host threads,
include/cuadmm/eig_cpu.h:31-51, src/duo_solver.cu:344-371,598-606) as a timed CPU baseline.

TEST / BENCH INFRASTRUCTURE ONLY: imported by bench.py's cpu_baseline leg and by tests/, never by cuadmm_amd/.
"""
import ctypes as C
import glob
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libcpu_eig_baseline.so")
_lib = None


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    src = os.path.join(HERE, "cpu_eig_baseline.c")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-march=native", "-shared", "-fPIC", src, os.path.join(HERE, "eigproj_twin.c"),
                               "-o", SO, "-ldl", "-lpthread", "-lm"])
    return SO


def openblas_path():
    import scipy
    hits = sorted(glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs", "libscipy_openblas*.so*")))
    hits = [h for h in hits if "openblas64" not in os.path.basename(h)]      # LP64 build: 32-bit LAPACK integers
    if not hits:
        raise RuntimeError("no LP64 OpenBLAS bundled with scipy found")
    return os.path.abspath(hits[0])


def load():
    global _lib
    if _lib is None:
        lib = C.CDLL(build())
        lib.cpu_eig_init.restype = C.c_int
        lib.cpu_eig_init.argtypes = [C.c_char_p]
        lib.cpu_psd_project.restype = C.c_double
        lib.cpu_psd_project.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        rc = lib.cpu_eig_init(openblas_path().encode())
        if rc != 0:
            raise RuntimeError("cpu_eig_init failed (%d)" % rc)
        _lib = lib
    return _lib


def psd_project(xb, blk, threads=1, eig_only=False, engine="lapack"):
    """-> (projected svec, seconds of the threaded region); engine: "lapack" (dsyevd + DGEMM) | "ql" (scalar twin)"""
    lib = load()
    if engine != "ql":
        threads = min(int(threads), 48)     # scipy's OpenBLAS is built for <= 64 caller threads and aborts beyond
    xb = np.ascontiguousarray(xb, dtype=np.float64)
    blk = np.ascontiguousarray(blk, dtype=np.int32)
    out = np.zeros_like(xb)
    secs = lib.cpu_psd_project(xb.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                               blk.ctypes.data_as(C.c_void_p), int(blk.size), int(threads), int(bool(eig_only)), 1 if engine == "ql" else 0)
    if secs < 0:
        raise RuntimeError("cpu_psd_project failed (%g)" % secs)
    return out, secs
