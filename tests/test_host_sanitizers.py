"""AddressSanitizer + UBSan over the host-side code (sanitizer target): the TXT loader, block bookkeeping, the
hand-written quotient-graph minimum-degree ordering + up-looking LDL^T (aat_ldlt.cpp), its split / threaded solves, the
spin-then-sleep host pool and the schedule model, driven by cuadmm_host_selftest (csrc/host_selftest.cpp) on two shipped
problems.  CPU only: the library is a separate build of the host sources (no HIP), run in a child process under
LD_PRELOAD=libasan (sanitizers are not available on the GPU pool)."""
import os
import subprocess
import sys
import textwrap

import pytest

from cuadmm_amd import build as b


@pytest.mark.parametrize("name", ["hinf12", "PushT_N=10_MOMENT"])
def test_host_sources_clean_under_asan_ubsan(name, problem_dirs, tmp_path):
    lib = b.build_host_sanitized()
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan not available")
    code = textwrap.dedent("""
        import ctypes, sys
        lib = ctypes.CDLL(sys.argv[1])
        lib.cuadmm_host_selftest.restype = ctypes.c_int
        rc = lib.cuadmm_host_selftest(sys.argv[2].encode(), sys.argv[3].encode())
        print("selftest rc", rc)
        sys.exit(1 if rc else 0)
    """)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               CUADMM_HOST_THREADS="4")
    r = subprocess.run([sys.executable, "-c", code, lib, problem_dirs[name], str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "selftest rc 0" in r.stdout
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
