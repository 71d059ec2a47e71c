// Explicit eigendecomposition of one LARGE dense symmetric matrix on the whole chip (eig_large.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace cuadmm {

constexpr int kEigLargeMin = 129;    // below: one workgroup per matrix with the matrix in LDS (psd_wg_kernel) is faster
constexpr int kEigLargeMax = 8192;   // d and e^2 of the tridiagonal matrix live in LDS during the bisection (2 n doubles)

// mat: n x n column-major symmetric (device), overwritten by the eigenvectors (column j <-> W[j]); W: ascending eigenvalues;
// info (device, may be null): 0, or 1 when the orthonormalisation of the inverse-iteration vectors did not converge.
// Synchronous with respect to `st` (allocates and frees its workspace).
int eig_large(double* mat, double* W, int* info, int n, hipStream_t st);

}  // namespace cuadmm
