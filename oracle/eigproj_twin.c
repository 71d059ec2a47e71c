/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
 *
 * Scalar CPU twin of the arithmetic inside the HIP projection kernels
 * (cuadmm_amd/csrc/psd_kernels.hip): the per-block step of the reference's hot loop
 *     svec -> smat unpack        src/kernels/vec_mat_conversion.cu:11-34
 *     symmetric eigendecomposition   include/cuadmm/cusolver.h:76-95,154-171 (cuSOLVER there)
 *     W = max(W,0)               src/kernels/dense_scalar.cu:41-47
 *     P = V diag(W) V^T          src/kernels/diagonal_batch.cu:11-23 + include/cuadmm/cublas.h:18-35
 *     smat -> svec pack          src/kernels/vec_mat_conversion.cu:36-57
 * with the eigendecomposition done the way the kernels do it: Householder tridiagonalisation
 * (reflectors kept in the lower triangle, Q formed in place by backward accumulation) followed
 * by implicit-shift QL with Wilkinson shifts (the textbook tql2/imtql2 recurrence).
 *
 * Pinned by tests/test_oracle_pinning.py against LAPACK dsyevd (numpy.linalg.eigh), which is the
 * reference's own eig_cpu routine (include/cuadmm/eig_cpu.h:31-51), and against the closed-form
 * spectra hard-coded in the reference tests (test/cusolver_test.hpp:60-63,178-182).
 *
 * Build: gcc -O2 -shared -fPIC oracle/eigproj_twin.c -o oracle/_build/libeigproj_twin.so -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static const double SQRT2 = 0x1.6a09e667f3bccp+0;    /* include/cuadmm/kernels.h:180 */
static const double SQRT2INV = 0x1.6a09e667f3bcdp-1; /* include/cuadmm/kernels.h:181 */

/* M: n x n row-major, leading dimension ld, full symmetric on entry.
 * On exit M = Z (Z[r][k] = component r of eigenvector k), d = eigenvalues (unsorted).
 * e, tau, vv, ww: workspaces of n doubles.  Returns 0, or 1 if QL hit its sweep cap. */
int twin_sym_eig(double* M, int ld, int n, double* d, double* e, double* tau, double* vv, double* ww) {
  int k, r, c;
  /* ---- Householder tridiagonalisation ---- */
  for (k = 0; k < n - 2; ++k) {
    double alpha = M[(k + 1) * ld + k];
    double xn2 = 0.0;
    for (r = k + 2; r < n; ++r) xn2 += M[r * ld + k] * M[r * ld + k];
    double t, beta;
    if (xn2 == 0.0) {
      t = 0.0; beta = alpha;
    } else {
      beta = -copysign(sqrt(alpha * alpha + xn2), alpha);
      t = (beta - alpha) / beta;
      double scal = 1.0 / (alpha - beta);
      for (r = k + 2; r < n; ++r) { M[r * ld + k] *= scal; vv[r] = M[r * ld + k]; }
      vv[k + 1] = 1.0;
    }
    e[k] = beta; tau[k] = t;
    if (t != 0.0) {
      double K = 0.0;
      for (r = k + 1; r < n; ++r) {
        double p = 0.0;
        for (c = k + 1; c < n; ++c) p += M[r * ld + c] * vv[c];
        p *= t;
        ww[r] = p;
        K += p * vv[r];
      }
      K *= -0.5 * t;
      for (r = k + 1; r < n; ++r) ww[r] += K * vv[r];
      for (r = k + 1; r < n; ++r)
        for (c = k + 1; c < n; ++c) M[r * ld + c] -= vv[r] * ww[c] + ww[r] * vv[c];
    }
  }
  for (r = 0; r < n; ++r) d[r] = M[r * ld + r];
  if (n >= 2) e[n - 2] = M[(n - 1) * ld + (n - 2)];
  e[n - 1] = 0.0;
  /* ---- form Q in place (backward accumulation) ---- */
  M[(n - 1) * ld + (n - 1)] = 1.0;
  for (k = n - 3; k >= 0; --k) {
    double t = tau[k];
    vv[k + 1] = 1.0;
    for (r = k + 2; r < n; ++r) vv[r] = M[r * ld + k];
    M[(k + 1) * ld + (k + 1)] = 1.0;
    for (c = k + 2; c < n; ++c) { M[(k + 1) * ld + c] = 0.0; M[c * ld + (k + 1)] = 0.0; }
    if (t != 0.0) {
      for (c = k + 1; c < n; ++c) {
        double s = 0.0;
        for (r = k + 1; r < n; ++r) s += vv[r] * M[r * ld + c];
        s *= t;
        for (r = k + 1; r < n; ++r) M[r * ld + c] -= vv[r] * s;
      }
    }
  }
  if (n >= 2) {
    M[0] = 1.0;
    for (c = 1; c < n; ++c) { M[c] = 0.0; M[c * ld] = 0.0; }
  }
  /* ---- implicit QL ---- */
  int fail = 0;
  double tn = 0.0; /* deflation threshold relative to ||T||_F (see psd_device.h) */
  for (r = 0; r < n; ++r) tn += d[r] * d[r] + 2.0 * e[r] * e[r];
  const double eps_abs = sqrt(tn) * 0x1p-53;
  for (int l = 0; l < n; ++l) {
    int iter = 0, m;
    do {
      for (m = l; m < n - 1; ++m) {
        double dd = fabs(d[m]) + fabs(d[m + 1]);
        if (fabs(e[m]) <= eps_abs || fabs(e[m]) + dd == dd) break;
      }
      if (m != l) {
        if (iter++ == 60) { fail = 1; break; }
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double rr = sqrt(g * g + 1.0);
        g = d[m] - d[l] + e[l] / (g + copysign(rr, g));
        double s = 1.0, cc = 1.0, p = 0.0;
        int i;
        for (i = m - 1; i >= l; --i) {
          double f = s * e[i], b = cc * e[i];
          rr = sqrt(f * f + g * g);
          e[i + 1] = rr;
          if (rr == 0.0) { d[i + 1] -= p; e[m] = 0.0; break; }
          s = f / rr; cc = g / rr;
          g = d[i + 1] - p;
          rr = (d[i] - g) * s + 2.0 * cc * b;
          p = s * rr;
          d[i + 1] = g + p;
          g = cc * rr - b;
          for (r = 0; r < n; ++r) {
            double z1 = M[r * ld + i + 1], z0 = M[r * ld + i];
            M[r * ld + i + 1] = s * z0 + cc * z1;
            M[r * ld + i] = cc * z0 - s * z1;
          }
        }
        if (rr == 0.0 && i >= l) continue;
        d[l] -= p; e[l] = g; e[m] = 0.0;
      }
    } while (m != l);
  }
  return fail;
}

/* Projection of one block given in svec form (length n(n+1)/2) onto the PSD cone. */
int twin_psd_project_block(const double* xin, double* xout, int n) {
  int ld = n | 1;
  double* M = (double*)malloc(sizeof(double) * (size_t)(n * ld + 5 * n));
  double *d = M + n * ld, *e = d + n, *tau = e + n, *vv = tau + n, *ww = vv + n;
  int idx = 0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j, ++idx) {
      double v = xin[idx] * (i == j ? 1.0 : SQRT2INV);
      M[j * ld + i] = v; M[i * ld + j] = v;
    }
  int fail = twin_sym_eig(M, ld, n, d, e, tau, vv, ww);
  for (int k = 0; k < n; ++k) vv[k] = d[k] > 0.0 ? d[k] : 0.0;
  idx = 0;
  for (int b = 0; b < n; ++b)
    for (int a = 0; a <= b; ++a, ++idx) {
      double acc = 0.0;
      for (int k = 0; k < n; ++k) acc += (M[a * ld + k] * vv[k]) * M[b * ld + k];
      xout[idx] = acc * (a == b ? 1.0 : SQRT2);
    }
  free(M);
  return fail;
}

/* svec of many blocks: blk[nb] sizes in order; returns number of blocks whose QL failed. */
int twin_psd_project(const double* xin, double* xout, const int* blk, int nb) {
  long long off = 0;
  int fails = 0;
  for (int k = 0; k < nb; ++k) {
    fails += twin_psd_project_block(xin + off, xout + off, blk[k]);
    off += (long long)blk[k] * (blk[k] + 1) / 2;
  }
  return fails;
}

/* Dense symmetric eig: A n x n column-major (lower triangle read), overwritten by eigenvectors
 * (column k <-> W[k]), W ascending -- the contract of cusolver.h:76-95 / eig_cpu.h:31-51. */
int twin_eig_dense(double* A, double* W, int n) {
  int ld = n | 1;
  double* M = (double*)malloc(sizeof(double) * (size_t)(n * ld + 5 * n));
  double *d = M + n * ld, *e = d + n, *tau = e + n, *vv = tau + n, *ww = vv + n;
  for (int c = 0; c < n; ++c)
    for (int r = c; r < n; ++r) { M[r * ld + c] = A[c * n + r]; M[c * ld + r] = A[c * n + r]; }
  int fail = twin_sym_eig(M, ld, n, d, e, tau, vv, ww);
  for (int k = 0; k < n; ++k) {
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += (d[j] < d[k]) || (d[j] == d[k] && j < k);
    W[rank] = d[k];
    for (int r = 0; r < n; ++r) A[rank * n + r] = M[r * ld + k];
  }
  free(M);
  return fail;
}
