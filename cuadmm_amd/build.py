"""Builds the gfx950 shared library (HIP kernels + C ABI) and the cuadmm_exe front end in-tree.

    python -m cuadmm_amd.build [--force]

Outputs (git-ignored, but shipped to the GPU box with the working tree):
    cuadmm_amd/lib/libcuadmm_amd.so
    cuadmm_amd/lib/cuadmm_exe
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
INCLUDE = os.path.join(HERE, "..", "include")
LIB = os.path.join(LIBDIR, "libcuadmm_amd.so")
EXE = os.path.join(LIBDIR, "cuadmm_exe")

HIP_SOURCES = ["psd_kernels.hip", "psd_large.hip", "tail_solve.hip", "eig_large.hip", "lead_solve.hip", "vec_kernels.hip", "staging.hip", "engine.hip", "duo_group.hip"]
CPP_SOURCES = ["io.cpp", "blocks.cpp", "aat_ldlt.cpp", "mex_entry.cpp"]
HEADERS = None   # every header under csrc/ plus the public C header (computed in build())
ARCH = "gfx950"


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (needed to build the gfx950 kernels)")


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def build(force=False, verbose=False):
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")]
    headers.append(os.path.join(INCLUDE, "cuadmm_amd.h"))
    common = (["-DCUADMM_QL_CHECKS"] if os.environ.get("CUADMM_QL_CHECKS") else []) + ["-O3", "-std=c++17", "-fPIC", "-pthread", "-I" + INCLUDE, "-I" + CSRC, "-Wno-unused-result"]
    jobs = []
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src + ".o")
        if force or not _newer(o, [s] + headers):
            jobs.append([hipcc, "--offload-arch=" + ARCH] + common + ["-c", s, "-o", o])
    for src in CPP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src + ".o")
        if force or not _newer(o, [s] + headers):
            jobs.append(["g++"] + common + ["-c", s, "-o", o])
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        for out in ex.map(_run, jobs):
            if verbose and out:
                print(out)
    objs = [os.path.join(OBJDIR, s + ".o") for s in HIP_SOURCES + CPP_SOURCES]
    if force or jobs or not os.path.exists(LIB):
        _run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs +
             ["-ldl", "-pthread", "-Wl,-rpath,/opt/rocm/lib"])
    cli = os.path.join(CSRC, "cli_main.cpp")
    if force or jobs or not _newer(EXE, [cli, LIB]):
        _run(["g++", "-O2", "-std=c++17", "-I" + INCLUDE, cli, "-o", EXE, "-L" + LIBDIR, "-lcuadmm_amd",
              "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"])
    return LIB


def build_host_sanitized():
    """ASan + UBSan build of the HOST sources only (loader, block bookkeeping, A*A^T ordering / factor / solve, schedule
    model) -> cuadmm_amd/lib/libcuadmm_host_asan.so.  CPU only (the GPU pool refuses sanitizers); driven by
    tests/test_host_sanitizers.py through tests/_asan_host_driver.py under LD_PRELOAD=libasan."""
    os.makedirs(LIBDIR, exist_ok=True)
    out = os.path.join(LIBDIR, "libcuadmm_host_asan.so")
    srcs = [os.path.join(CSRC, s) for s in ("io.cpp", "blocks.cpp", "aat_ldlt.cpp", "host_selftest.cpp")]
    headers = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")]
    if not _newer(out, srcs + headers):
        _run(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-pthread", "-fsanitize=address,undefined",
              "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-I" + INCLUDE, "-I" + CSRC] + srcs + ["-o", out])
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
