// Dense tail of the A A^T solve on the GPU (tail_solve.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace cuadmm {

struct TailSolve {
  int k = 0, K = 0;            // tail size, padded to a multiple of 64
  double *W = nullptr, *Wt = nullptr, *dinv = nullptr, *vin = nullptr, *vmid = nullptr;   // device
  double* h_vec = nullptr;     // pinned staging vector
  double build_s = 0, factor_s = 0;
  int build(const double* L22, const double* D2, int k, hipStream_t st);                                        // host factor
  int build_from_schur(const long long* row_ptr, const int* col, const double* val, int k, hipStream_t st);    // GPU factor
  int alloc(int k);
  int invert(const double* dL, hipStream_t st);
  int solve(double* z2, hipStream_t st);
  int solve_device(hipStream_t st);      // in place on the device vector `vin` (k entries, zero padding kept); asynchronous
  void release();
  ~TailSolve() { release(); }
};

}  // namespace cuadmm
