// Explicit eigendecomposition of one LARGE dense symmetric matrix on the whole chip (eig_large.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

namespace cuadmm {

constexpr int kEigLargeMin = 129;    // below: one workgroup per matrix with the matrix in LDS (psd_wg_kernel) is faster
constexpr int kEigLargeMax = 8192;   // d and e^2 of the tridiagonal matrix live in LDS during the bisection (2 n doubles)

// Device workspace of the path (~17 n^2 doubles), kept by whoever calls it repeatedly (the projection plan: every iteration)
struct EigLargeWs {
  int cap = 0;                 // largest n the buffers hold
  std::vector<void*> bufs;
  double *H = nullptr, *tau = nullptr, *dvec = nullptr, *evec = nullptr, *P = nullptr, *lamp = nullptr, *scal = nullptr, *work = nullptr,
         *Z = nullptr, *M = nullptr, *Mt = nullptr, *G = nullptr, *Winv = nullptr, *Wtmp = nullptr, *dd = nullptr, *Yp = nullptr, *err = nullptr,
         *dense = nullptr, *Wd = nullptr;     // dense / Wd: the matrix and eigenvalues of eig_large_project
  int* dflag = nullptr;
  int ensure(int n);
  void release();
  EigLargeWs() = default;
  EigLargeWs(const EigLargeWs&) = delete;              // owns device memory
  EigLargeWs& operator=(const EigLargeWs&) = delete;
  ~EigLargeWs() { release(); }
};

// mat: n x n column-major symmetric (device), overwritten by the eigenvectors (column j <-> W[j]); W: ascending eigenvalues;
// info (device, may be null): 0, or 1 when the orthonormalisation of the inverse-iteration vectors did not converge.
// Synchronous with respect to `st`.  ws == nullptr: a workspace is allocated and freed inside.
int eig_large(double* mat, double* W, int* info, int n, hipStream_t st, EigLargeWs* ws = nullptr);

// Projection of ONE block through the explicit eigendecomposition: svec in -> svec out of sum_k max(lambda_k, 0) v_k v_k^T over the
// eig_rank largest eigenvalues (eig_rank <= 0: all) -- the rank-limited projection of the reference (src/kernels/dense_scalar.cu:
// 41-57, src/utils/get_eig_rank_mask.cu:13-37) for blocks beyond the one-workgroup kernels.  fail (device, may be null) is
// incremented when the eigendecomposition reports info != 0.
int eig_large_project(const double* svec_in, double* svec_out, int n, int eig_rank, int* fail, hipStream_t st, EigLargeWs* ws);

}  // namespace cuadmm
