"""A main.cu-shaped caller (reference src/main.cu:8-44: Problem::from_txt, SDPSolver::init with the 20 positional arguments,
solve(1e6, 1e-3, false, 50, 100, 5000), X.to_txt) compiled against include/cuadmm_amd.hpp and linked with libcuadmm_amd.so.
Without a GPU the program must fail LOUDLY (there is no CPU fallback); on the GPU box it must write X_opt.txt and print the
reference's console table."""
import os
import subprocess

import numpy as np
import pytest

import cuadmm_amd
from tests.conftest import ROOT

CALLER = r'''
#include <iostream>
#include <string>
#include <tuple>
#include <vector>

#include "cuadmm_amd.hpp"
using namespace cuadmm_amd;

int main(int argc, char* argv[]) {
  if (argc < 2) return 2;
  std::string prefix = argv[1];
  const int eig_stream_num_per_gpu = 15, cpu_eig_thread_num = 30;
  try {
    Problem problem;
    problem.from_txt(prefix);
    std::vector<int> blk_vals;
    for (const auto& b : problem.blk_vals) blk_vals.push_back(std::get<1>(b));
    SDPSolver solver;
    solver.set_option("verbose", argc > 2 ? 0.0 : 1.0);
    const double sig = 1e0;
    solver.init(eig_stream_num_per_gpu, cpu_eig_thread_num, problem.vec_len, problem.con_num, problem.At_csc_col_ptrs.data(),
                problem.At_csc_row_ids.data(), problem.At_csc_vals.data(), problem.At_nnz, problem.b_indices.data(),
                problem.b_vals.data(), problem.b_nnz, problem.C_indices.data(), problem.C_vals.data(), problem.C_nnz,
                blk_vals.data(), problem.mat_num, problem.X_vals.data(), problem.y_vals.data(), problem.S_vals.data(), sig);
    solver.solve((int)1e6, 1e-3, false, 50, 100, 5000);
    solver.X.to_txt(prefix + "X_opt.txt");
    std::cout << "iterations " << solver.info_iter_num << " X " << solver.X.size << std::endl;
  } catch (const std::exception& e) {
    std::cerr << "cuadmm_amd: " << e.what() << std::endl;
    return 3;
  }
  return 0;
}
'''


def _build(tmp_path):
    cuadmm_amd.load()
    src = tmp_path / "caller.cpp"
    src.write_text(CALLER)
    exe = tmp_path / "caller"
    libdir = os.path.join(ROOT, "cuadmm_amd", "lib")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L" + libdir, "-lcuadmm_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"])
    return str(exe)


def test_reference_shaped_caller_compiles_links_and_fails_loudly_without_a_device(tmp_path, problem_dirs):
    exe = _build(tmp_path)
    if cuadmm_amd.load().cuadmm_device_count() > 0:
        pytest.skip("GPU present: covered by the gpu test below")
    r = subprocess.run([exe, problem_dirs["hinf12"]], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3
    assert "no HIP device" in r.stderr and "no CPU fallback" in r.stderr
    assert not os.path.exists(os.path.join(problem_dirs["hinf12"], "X_opt.txt"))


@pytest.mark.gpu
def test_reference_shaped_caller_runs_on_the_gpu(tmp_path, problem_dirs):
    exe = _build(tmp_path)
    d = problem_dirs["truss5"]
    r = subprocess.run([exe, d], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "  it. | p infeas d infeas | primal obj.   dual obj. rel. gap |  time |   sigma | " in r.stdout
    x = np.loadtxt(os.path.join(d, "X_opt.txt"))
    p = cuadmm_amd.Problem.from_txt(d)
    assert x.size == p.vec_len and np.all(np.isfinite(x))
