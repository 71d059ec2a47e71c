// cuadmm_exe <dir/> : the reference's command line front end (src/main.cu:8-44) on top of the C ABI.
//
// Same positional form, same constants: eig_streams=15, cpu_eig_threads=30, sig=1.0 and
// solve(1e6, 1e-3, /*sig_update_threshold=*/false -> 0, 50, 100, 5000) (main.cu:10-11,23,39), reads
// <dir/>{blk,con_num,At,b,C}.txt and writes <dir/>X_opt.txt with "%.32f" per line (memory.h:278-294).
// Optional trailing --key=value arguments (not in the reference) override the solve parameters:
//   --max_iter= --stop_tol= --threshold= --stage1= --stage2= --switch_admm= --sigscale= --sig= --device= --quiet
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "cuadmm_amd.h"

static bool opt(const char* arg, const char* key, double& out) {
  size_t n = strlen(key);
  if (strncmp(arg, key, n) == 0 && arg[n] == '=') { out = atof(arg + n + 1); return true; }
  return false;
}

int main(int argc, char* argv[]) {
  if (argc < 2) {
    std::cerr << "usage: cuadmm_exe <problem_dir/> [--key=value ...]" << std::endl;
    return 1;
  }
  std::string prefix = argv[1];
  int eig_stream_num_per_gpu = 15, cpu_eig_thread_num = 30;
  double max_iter = 1e6, stop_tol = 1e-3, threshold = 0, stage1 = 50, stage2 = 100, switch_admm = 5000, sigscale = 1.05,
         sig = 1e0, device = 0;
  bool quiet = false;
  for (int i = 2; i < argc; ++i) {
    if (opt(argv[i], "--max_iter", max_iter) || opt(argv[i], "--stop_tol", stop_tol) || opt(argv[i], "--threshold", threshold) ||
        opt(argv[i], "--stage1", stage1) || opt(argv[i], "--stage2", stage2) || opt(argv[i], "--switch_admm", switch_admm) ||
        opt(argv[i], "--sigscale", sigscale) || opt(argv[i], "--sig", sig) || opt(argv[i], "--device", device))
      continue;
    if (strcmp(argv[i], "--quiet") == 0) { quiet = true; continue; }
    std::cerr << "unknown option " << argv[i] << std::endl;
    return 1;
  }

  cuadmm_problem* prob = nullptr;
  if (cuadmm_problem_from_txt(prefix.c_str(), &prob) != CUADMM_OK) {
    std::cerr << cuadmm_last_error() << std::endl;
    return 1;   // the reference exit(1)s on unreadable input (io.cu:30-33, problem.cu:33-35)
  }
  cuadmm_problem_view v;
  cuadmm_problem_view_get(prob, &v);

  cuadmm_solver* solver = nullptr;
  cuadmm_create(&solver);
  cuadmm_set_option(solver, "device", device);
  cuadmm_set_option(solver, "verbose", quiet ? 0 : 1);
  int rc = cuadmm_init(solver, eig_stream_num_per_gpu, cpu_eig_thread_num, v.vec_len, v.con_num, v.At_csc_col_ptrs,
                       v.At_csc_row_ids, v.At_csc_vals, v.At_nnz, v.b_indices, v.b_vals, v.b_nnz, v.C_indices, v.C_vals,
                       v.C_nnz, v.blk_vals, v.mat_num, nullptr, nullptr, nullptr, sig);
  if (rc != CUADMM_OK) {
    std::cerr << cuadmm_last_error() << std::endl;
    return 1;
  }
  rc = cuadmm_solve(solver, (int)max_iter, stop_tol, (int)threshold, (int)stage1, (int)stage2, (int)switch_admm, sigscale, 1);
  if (rc != CUADMM_OK) std::cerr << cuadmm_last_error() << std::endl;

  std::vector<double> X((size_t)v.vec_len);
  if (cuadmm_get_X(solver, X.data()) == CUADMM_OK) cuadmm_write_dense_txt((prefix + "X_opt.txt").c_str(), X.data(), v.vec_len);
  cuadmm_destroy(solver);
  cuadmm_problem_free(prob);
  return rc == CUADMM_OK ? 0 : 1;
}
