// PSD projection of blocks with n > 64 through the matrix sign function on the fp64 matrix cores (psd_large.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

namespace cuadmm {

struct SignPsd {
  static constexpr int kLiftSteps = 36;      // scaled Newton-Schulz steps (mu = kLiftMu): resolves |lambda| >= 1e-13 ||X||_1
  static constexpr int kPolishSteps = 8;     // plain steps: quadratic convergence from [0.5, 1]
  static constexpr double kLiftMu = 1.53;    // p(mu) = 0.5: converged eigenvalues never drop below 0.5
  struct Group { int N = 0, begin = 0, count = 0; };
  std::vector<Group> groups;                 // same padded size N, bounded workspace
  int* d_ids = nullptr;                      // block ids, group after group
  int* d_steps = nullptr;                    // not owned; when set: Newton-Schulz steps taken per block
  double *X0 = nullptr, *S = nullptr, *Y = nullptr, *T = nullptr, *colsum = nullptr, *scale = nullptr;
  int build(const int* blk, const std::vector<int>& members);
  void release();
  int project(const double* in, double* out, const long long* boff, const int* bn, int* d_fail, hipStream_t st);
  int project_launch(const double* in, double* out, const long long* boff, const int* bn, int* d_fail, hipStream_t st);
  bool allow_graph = false;                  // set by long-lived owners (the engine); one-shot plans launch directly
  hipGraphExec_t graph_exec = nullptr;       // captured launch sequence of project_launch for (g_in, g_out, ...)
  const double* g_in = nullptr; double* g_out = nullptr; const long long* g_boff = nullptr; const int* g_bn = nullptr; int* g_fail = nullptr;
  bool empty() const { return groups.empty(); }
  ~SignPsd() { release(); }
};

// C = alpha * A*B + beta * E, n x n row-major, A symmetric, n a multiple of 64 (E may be null)
int large_gemm_sym(int N, const double* A, const double* B, double alpha, double beta, const double* E, double* C, hipStream_t st);

}  // namespace cuadmm
