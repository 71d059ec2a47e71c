// Header-only C++ facade with the reference's class/method/member names on top of the C ABI
// (include/cuadmm_amd.h), so that callers written against the reference's SDPSolver
// (include/cuadmm/solver.h:30-248; src/main.cu:21-41) compile against this engine unchanged apart
// from the include line.  Differences: X/y/S are host copies fetched on demand (the reference exposes
// device pointers through DeviceDenseVector::vals), and failures throw std::runtime_error instead of
// printing and continuing (include/cuadmm/check.h:17-56).
#pragma once
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "cuadmm_amd.h"

namespace cuadmm_amd {

// Problem (include/cuadmm/problem.h:12-41; Problem::from_txt, src/problem.cu:11-83): the TXT directory of cuadmm_exe with the
// reference's member names -- blk_vals as (type, size) pairs, At in CSC over the constraints, sparse b and C; X / y / S stay
// empty (cold start: their data() is what main.cu hands to init).
class Problem {
 public:
  int vec_len = 0, con_num = 0, mat_num = 0, At_nnz = 0, b_nnz = 0, C_nnz = 0;
  std::vector<std::tuple<char, int>> blk_vals;
  std::vector<int> At_csc_col_ptrs, At_csc_row_ids, b_indices, C_indices;
  std::vector<double> At_csc_vals, b_vals, C_vals, X_vals, y_vals, S_vals;

  void from_txt(const std::string& prefix) {
    cuadmm_problem* p = nullptr;
    if (cuadmm_problem_from_txt(prefix.c_str(), &p) != CUADMM_OK) throw std::runtime_error(cuadmm_last_error());
    cuadmm_problem_view v;
    if (cuadmm_problem_view_get(p, &v) != CUADMM_OK) { cuadmm_problem_free(p); throw std::runtime_error(cuadmm_last_error()); }
    vec_len = v.vec_len; con_num = v.con_num; mat_num = v.mat_num; At_nnz = v.At_nnz; b_nnz = v.b_nnz; C_nnz = v.C_nnz;
    At_csc_col_ptrs.assign(v.At_csc_col_ptrs, v.At_csc_col_ptrs + v.con_num + 1);
    At_csc_row_ids.assign(v.At_csc_row_ids, v.At_csc_row_ids + v.At_nnz);
    At_csc_vals.assign(v.At_csc_vals, v.At_csc_vals + v.At_nnz);
    b_indices.assign(v.b_indices, v.b_indices + v.b_nnz); b_vals.assign(v.b_vals, v.b_vals + v.b_nnz);
    C_indices.assign(v.C_indices, v.C_indices + v.C_nnz); C_vals.assign(v.C_vals, v.C_vals + v.C_nnz);
    blk_vals.clear();
    for (int k = 0; k < v.mat_num; ++k) blk_vals.emplace_back(v.blk_vals[k] < 0 ? 'u' : 's', v.blk_vals[k] < 0 ? -v.blk_vals[k] : v.blk_vals[k]);
    cuadmm_problem_free(p);
  }
};

class SDPSolver {
 public:
  // results, named as in the reference (solver.h:149-161)
  int info_iter_num = 0;
  std::vector<double> info_pobj_arr, info_dobj_arr, info_errRp_arr, info_errRd_arr, info_relgap_arr, info_sig_arr,
      info_bscale_arr, info_Cscale_arr;
  double total_time = 0.0;
  int vec_len = 0, con_num = 0;

  struct HostVector {  // stands in for DeviceDenseVector<double>: .vals / .size, plus to_txt (memory.h:278-294)
    std::vector<double> data;
    double* vals = nullptr;
    int size = 0;
    void to_txt(const std::string& filename) const {
      if (cuadmm_write_dense_txt(filename.c_str(), data.data(), (int64_t)data.size()) != CUADMM_OK)
        throw std::runtime_error(cuadmm_last_error());
    }
  };
  HostVector X, y, S;

  SDPSolver() { check(cuadmm_create(&h_)); }
  ~SDPSolver() { cuadmm_destroy(h_); }
  SDPSolver(const SDPSolver&) = delete;
  SDPSolver& operator=(const SDPSolver&) = delete;

  void set_option(const char* key, double value) { check(cuadmm_set_option(h_, key, value)); }
  cuadmm_solver* handle() { return h_; }

  // SDPSolver::init, same argument list and defaults (solver.h:208-223)
  void init(int eig_stream_num_per_gpu, int cpu_eig_thread_num, int vec_len_, int con_num_, int* cpu_At_csc_col_ptrs,
            int* cpu_At_csc_row_ids, double* cpu_At_csc_vals, int At_nnz, int* cpu_b_indices, double* cpu_b_vals, int b_nnz,
            int* cpu_C_indices, double* cpu_C_vals, int C_nnz, int* cpu_blk_vals, int mat_num, double* cpu_X_vals = nullptr,
            double* cpu_y_vals = nullptr, double* cpu_S_vals = nullptr, double sig = 1.0) {
    check(cuadmm_init(h_, eig_stream_num_per_gpu, cpu_eig_thread_num, vec_len_, con_num_, cpu_At_csc_col_ptrs,
                      cpu_At_csc_row_ids, cpu_At_csc_vals, At_nnz, cpu_b_indices, cpu_b_vals, b_nnz, cpu_C_indices, cpu_C_vals,
                      C_nnz, cpu_blk_vals, mat_num, cpu_X_vals, cpu_y_vals, cpu_S_vals, sig));
    vec_len = vec_len_;
    con_num = con_num_;
  }

  // SDPDuoSolver::init (duo_solver.h:236-255): two leading arguments more, default sig = 2e2
  void duo_init(bool if_gpu_eig_mom, int device_num_requested, int eig_stream_num_per_gpu, int cpu_eig_thread_num, int vec_len_,
                int con_num_, int* cpu_At_csc_col_ptrs, int* cpu_At_csc_row_ids, double* cpu_At_csc_vals, int At_nnz,
                int* cpu_b_indices, double* cpu_b_vals, int b_nnz, int* cpu_C_indices, double* cpu_C_vals, int C_nnz,
                int* cpu_blk_vals, int mat_num, double* cpu_X_vals = nullptr, double* cpu_y_vals = nullptr,
                double* cpu_S_vals = nullptr, double sig = 2e2) {
    check(cuadmm_duo_init(h_, if_gpu_eig_mom ? 1 : 0, device_num_requested, eig_stream_num_per_gpu, cpu_eig_thread_num, vec_len_,
                          con_num_, cpu_At_csc_col_ptrs, cpu_At_csc_row_ids, cpu_At_csc_vals, At_nnz, cpu_b_indices, cpu_b_vals,
                          b_nnz, cpu_C_indices, cpu_C_vals, C_nnz, cpu_blk_vals, mat_num, cpu_X_vals, cpu_y_vals, cpu_S_vals, sig));
    vec_len = vec_len_;
    con_num = con_num_;
  }

  // SDPSolver::solve, same argument list and defaults (solver.h:236-244)
  void solve(int max_iter, double stop_tol, int sig_update_threshold = 500, int sig_update_stage_1 = 50,
             int sig_update_stage_2 = 100, int switch_admm = (int)1.1e4, double sigscale = 1.05, bool if_first = true) {
    int rc = cuadmm_solve(h_, max_iter, stop_tol, sig_update_threshold, sig_update_stage_1, sig_update_stage_2, switch_admm,
                          sigscale, if_first ? 1 : 0);
    fetch();
    check(rc);
  }

 private:
  cuadmm_solver* h_ = nullptr;
  static void check(int rc) {
    if (rc < 0) throw std::runtime_error(cuadmm_last_error());
  }
  void fetch_vec(HostVector& v, int n, int (*get)(cuadmm_solver*, double*)) {
    v.data.resize((size_t)n);
    check(get(h_, v.data.data()));
    v.vals = v.data.data();
    v.size = n;
  }
  void fetch() {
    int64_t b = 0, e = 0;
    check(cuadmm_get_shard(h_, &b, &e, nullptr, nullptr));
    fetch_vec(X, (int)(e - b), cuadmm_get_X);
    fetch_vec(S, (int)(e - b), cuadmm_get_S);
    fetch_vec(y, con_num, cuadmm_get_y);
    info_iter_num = cuadmm_get_info_iter_num(h_);
    std::vector<double>* arrs[8] = {&info_pobj_arr, &info_dobj_arr, &info_errRp_arr, &info_errRd_arr,
                                    &info_relgap_arr, &info_sig_arr, &info_bscale_arr, &info_Cscale_arr};
    for (int k = 0; k < 8; ++k) {
      std::vector<double> tmp(1 << 20);
      int n = cuadmm_get_info_array(h_, k, tmp.data(), (int)tmp.size());
      if (n < 0) n = 0;
      arrs[k]->assign(tmp.begin(), tmp.begin() + n);
    }
    total_time = cuadmm_get_total_time(h_);
  }
};

}  // namespace cuadmm_amd
