// cuadmm_exe <dir/> : the reference's command line front end (src/main.cu:8-44) on top of the C ABI.
//
// Same positional form, same constants: eig_streams=15, cpu_eig_threads=30, sig=1.0 and
// solve(1e6, 1e-3, /*sig_update_threshold=*/false -> 0, 50, 100, 5000) (main.cu:10-11,23,39), reads
// <dir/>{blk,con_num,At,b,C}.txt and writes <dir/>X_opt.txt with "%.32f" per line (memory.h:278-294).
// Optional trailing --key=value arguments (not in the reference) override the solve parameters:
//   --max_iter= --stop_tol= --threshold= --stage1= --stage2= --switch_admm= --sigscale= --sig= --device= --quiet
//   --json=<file>: a sidecar with the run's figures (iterations, residuals, iters/s, per-phase milliseconds, the projection's
//   nominal TFLOP/s = 10.67 sum n^3 per projection and the vector kernels' algorithmic GB/s: SURVEY.md 8d); switches the
//   engine's per-phase HIP-event timers on (option "profile").  Nothing is written unless asked for: the reference writes X_opt.txt only.
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

// --json=<file>: one object; every number printed with %.17g
static bool write_sidecar(const std::string& path, const std::string& prefix, cuadmm_solver* solver, const cuadmm_problem_view& v) {
  FILE* f = fopen(path.c_str(), "w");
  if (!f) return false;
  const int iters = cuadmm_get_info_iter_num(solver);
  const double total_s = cuadmm_get_total_time(solver);
  double st[12] = {0}, prof[3 * CUADMM_NUM_KCLASS] = {0}, cnt[8] = {0};
  cuadmm_get_state(solver, st);
  cuadmm_get_profile(solver, prof);
  cuadmm_get_counters(solver, cnt);
  double sum_n3 = 0;
  for (int k = 0; k < v.mat_num; ++k) { const double n = v.blk_vals[k] > 0 ? v.blk_vals[k] : 0; sum_n3 += n * n * n; }
  static const char* kname[CUADMM_NUM_KCLASS] = {"aty_xb", "psd_project", "post_proj", "spmv_A", "copies", "host_solve", "allreduce", "solve_gpu_part"};
  fprintf(f, "{\n \"problem\": \"%s\",\n \"vec_len\": %d, \"con_num\": %d, \"mat_num\": %d, \"At_nnz\": %d,\n", prefix.c_str(), v.vec_len, v.con_num, v.mat_num, v.At_nnz);
  fprintf(f, " \"iterations\": %d, \"total_time_s\": %.17g, \"iters_per_s\": %.17g,\n", iters, total_s, total_s > 0 ? iters / total_s : 0.0);
  fprintf(f, " \"errRp\": %.17g, \"errRd\": %.17g, \"pobj\": %.17g, \"dobj\": %.17g, \"relgap\": %.17g, \"sig\": %.17g, \"bscale\": %.17g, \"Cscale\": %.17g,\n",
          st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7]);
  fprintf(f, " \"eig_not_converged\": %.17g,\n", st[11]);
  fprintf(f, " \"plan\": {\"fused\": %d, \"closed_blocks\": %d, \"device_solve\": %d, \"factor_gpu_tail\": %d, \"batched_launches\": %.0f, \"iterations_in_batches\": %.0f},\n",
          (int)cnt[4], (int)cnt[5], (int)cnt[6], (int)cnt[7], cnt[0], cnt[1]);
  fprintf(f, " \"phases\": {");
  for (int k = 0; k < CUADMM_NUM_KCLASS; ++k) {
    const double launches = prof[3 * k], ms = prof[3 * k + 1], bytes = prof[3 * k + 2];
    fprintf(f, "%s\n  \"%s\": {\"launches\": %.0f, \"ms\": %.17g, \"algorithmic_bytes_per_launch\": %.17g, \"gb_per_s\": %.17g}", k ? "," : "", kname[k], launches, ms,
            bytes, ms > 0 ? bytes * launches / ms * 1e-6 : 0.0);
  }
  // the projection: one launch group per iteration, 10.67 n^3 nominal flops per block
  const double pl = prof[3 * 1], pms = prof[3 * 1 + 1];
  fprintf(f, "\n },\n \"psd_project_nominal_tflops\": %.17g\n}\n", pms > 0 ? 32.0 / 3.0 * sum_n3 * pl / pms * 1e-9 : 0.0);
  return fclose(f) == 0;
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
  std::string json_path;
  for (int i = 2; i < argc; ++i) {
    if (strncmp(argv[i], "--json=", 7) == 0) { json_path = argv[i] + 7; continue; }
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
  if (!json_path.empty()) cuadmm_set_option(solver, "profile", 1);
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
  if (!json_path.empty() && !write_sidecar(json_path, prefix, solver, v)) std::cerr << "cannot write " << json_path << std::endl;
  cuadmm_destroy(solver);
  cuadmm_problem_free(prob);
  return rc == CUADMM_OK ? 0 : 1;
}
