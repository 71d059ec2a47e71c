// Explicit eigendecomposition of one LARGE dense symmetric matrix (129 <= n <= 8192) on the whole chip.
//
// Contract: the reference's single_eig_cusolver (include/cuadmm/cusolver.h:76-95: cusolverDnXsyevd, eigenvectors in place,
// eigenvalues ascending), called per large block at src/solver.cu:540-564.  The solver's own projection does not need it (matrix
// sign function, psd_large.hip); this serves cuadmm_op_batch_eig -- the eigenvalues are among the outputs north_star lists.
//
// One workgroup running Householder + QL on a matrix in HBM (psd_wg_kernel<.., GLOBAL>) needed 3.2 s at n = 1024 and 76 s at
// n = 2000.  Here every phase is spread over the chip, and nothing is sequential in n^2:
//
//   1. tridiagonalisation  Q^T A Q = T, unblocked Householder on the full symmetric matrix, TWO launches per column
//      (el_house_symv_kernel: reflector + p = tau A v, one wavefront per column of the trailing matrix;
//       el_rank2_kernel: A -= v w^T + w v^T); every workgroup recomputes the reflector / the scalar p^T v itself in a fixed
//      order (bit-identical in all of them), so there is no third launch and no atomics.  HBM / Infinity-Cache bound:
//      n^3 / 3 * 8 B * 3 passes = 64 GB at n = 2000.
//   2. eigenvalues of T by bisection on the Sturm count (LAPACK dstebz's recurrence with its pivmin guard), ONE THREAD PER
//      EIGENVALUE, d and e^2 in LDS: n independent chains of ~55 n divisions.
//   3. eigenvectors of T by inverse iteration, ONE THREAD PER EIGENVECTOR: LU with partial pivoting of T - lambda I and the
//      perturbed triangular solves of LAPACK dlagtf / dlagts(job = -1), start vector, scaling and stopping rule of dstein
//      (work arrays interleaved over the threads: coalesced).  dstein re-orthogonalises inside clusters by modified
//      Gram-Schmidt, one vector after the other -- O(n^3) sequential work when the clusters chain (n = 2000: every gap is below its
//      1e-3 ||T|| threshold).  Instead:
//   4. ALL vectors are orthonormalised at once by Cholesky-QR passes on the fp64 matrix cores: G = Z^T Z, G = L D L^T (dense
//      blocked LDL^T of tail_solve.hip), Z <- Z L^-T D^-1/2 with inv(L) by recursive doubling; repeated until
//      max |G - I| <= 1e-13 (two passes on separated spectra; a k-fold eigenvalue leaves k random vectors of its eigenspace, Gram
//      condition ~k^2: three).  A triangular mix only moves a vector inside the span of vectors it was not orthogonal to --
//      those with eigenvalues within ~eps ||T|| / (z_i^T z_j) of its own -- so residuals stay at eps ||T||.
//   5. back-transformation V = H_0 ... H_{n-2} Z: every workgroup keeps a few eigenvectors in LDS and applies all reflectors to
//      them (no dependency between workgroups: one launch).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <type_traits>
#include <vector>

#include "common.h"
#include "device_util.h"
#include "eig_large.h"
#include "tail_solve.h"
#include "wave_reduce.h"

namespace cuadmm {
namespace {

constexpr int EL_NT = 256;          // threads per workgroup of every kernel here
constexpr int EL_CPW = 4;           // trailing-matrix columns per workgroup (one per wavefront)
constexpr double EL_EPS = 1.1102230246251565e-16;      // LAPACK's eps = 2^-53 (dlamch('E'))
constexpr double EL_SAFMIN = 2.2250738585072014e-308;

// sum over the 256 threads of a workgroup in a fixed order; every thread receives it
__device__ __forceinline__ double el_block_sum(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- 1. tridiagonalisation -------------------------------------------------------------------------------------------
// Step k (LAPACK dsytd2, 'L'): x = A[k+1:n, k]; reflector H = I - tau v v^T with v[0] = 1, H x = beta e_1; p = tau A22 v.
// H (n x n, column-major) keeps v_k in rows k+1 .. n-1 of column k for the back-transformation.
__global__ __launch_bounds__(EL_NT) void el_house_symv_kernel(const double* __restrict__ A, long long ld, int n, int k,
                                                              double* __restrict__ H, double* __restrict__ tau,
                                                              double* __restrict__ dvec, double* __restrict__ evec, double* __restrict__ P) {
  extern __shared__ double el_sm[];
  double* v = el_sm;                 // t
  double* red = el_sm + (n - k - 1); // 4
  const int t = n - k - 1;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double* x = A + (size_t)k * ld + (k + 1);
  double ss = 0.0;
  for (int i = 1 + tid; i < t; i += EL_NT) { const double xi = x[i]; ss += xi * xi; }
  ss = el_block_sum(ss, red);
  const double alpha = x[0];
  double beta = alpha, tk = 0.0, scale = 0.0;
  if (ss > 0.0) {
    beta = -copysign(sqrt(alpha * alpha + ss), alpha);
    tk = (beta - alpha) / beta;
    scale = 1.0 / (alpha - beta);
  }
  for (int i = tid; i < t; i += EL_NT) v[i] = i == 0 ? 1.0 : x[i] * scale;
  __syncthreads();
  if (blockIdx.x == 0) {
    double* h = H + (size_t)k * ld + (k + 1);
    for (int i = tid; i < t; i += EL_NT) h[i] = v[i];
    if (tid == 0) { tau[k] = tk; evec[k] = beta; dvec[k] = A[(size_t)k * ld + k]; }
  }
#pragma unroll
  for (int q = 0; q < EL_CPW / 4; ++q) {
    const int j = (int)blockIdx.x * EL_CPW + wave * (EL_CPW / 4) + q;
    if (j >= t) continue;
    const double* col = A + (size_t)(k + 1 + j) * ld + (k + 1);      // row j of the symmetric trailing matrix
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;                  // four loads in flight per lane (fixed association: reproducible)
    int i = lane;
    for (; i + 192 < t; i += 256) {
      const double c0 = col[i], c1 = col[i + 64], c2 = col[i + 128], c3 = col[i + 192];
      s0 += c0 * v[i]; s1 += c1 * v[i + 64]; s2 += c2 * v[i + 128]; s3 += c3 * v[i + 192];
    }
    for (; i < t; i += 64) s0 += col[i] * v[i];
    const double s = wave_sum((s0 + s1) + (s2 + s3));
    if (lane == 0) P[j] = tk * s;
  }
}

// w = p - (tau / 2) (p^T v) v;  A22 -= v w^T + w v^T (the full square: both triangles stay valid, every access coalesced)
__global__ __launch_bounds__(EL_NT) void el_rank2_kernel(double* __restrict__ A, long long ld, int n, int k, const double* __restrict__ H,
                                                         const double* __restrict__ tau, const double* __restrict__ P) {
  extern __shared__ double el_sm[];
  const int t = n - k - 1;
  double* v = el_sm;           // t
  double* w = el_sm + t;       // t
  double* red = w + t;         // 4
  const double tk = tau[k];
  if (tk == 0.0) return;       // H = I (uniform over the launch)
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double* h = H + (size_t)k * ld + (k + 1);
  double g = 0.0;
  for (int i = tid; i < t; i += EL_NT) { const double vi = h[i], pi = P[i]; v[i] = vi; w[i] = pi; g += pi * vi; }
  g = el_block_sum(g, red);
  const double c = 0.5 * tk * g;
  for (int i = tid; i < t; i += EL_NT) w[i] -= c * v[i];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < EL_CPW / 4; ++q) {
    const int j = (int)blockIdx.x * EL_CPW + wave * (EL_CPW / 4) + q;
    if (j >= t) continue;
    double* col = A + (size_t)(k + 1 + j) * ld + (k + 1);
    const double vj = v[j], wj = w[j];
    int i = lane;
    for (; i + 192 < t; i += 256) {
      const double c0 = col[i], c1 = col[i + 64], c2 = col[i + 128], c3 = col[i + 192];
      col[i] = c0 - (v[i] * wj + w[i] * vj);
      col[i + 64] = c1 - (v[i + 64] * wj + w[i + 64] * vj);
      col[i + 128] = c2 - (v[i + 128] * wj + w[i + 128] * vj);
      col[i + 192] = c3 - (v[i + 192] * wj + w[i + 192] * vj);
    }
    for (; i < t; i += 64) col[i] -= v[i] * wj + w[i] * vj;
  }
}

__global__ void el_last_diag_kernel(const double* __restrict__ A, long long ld, int n, double* __restrict__ dvec) {
  if (threadIdx.x == 0 && blockIdx.x == 0) dvec[n - 1] = A[(size_t)(n - 1) * ld + (n - 1)];
}

// ---- 2. bisection ----------------------------------------------------------------------------------------------------
// number of eigenvalues of T that are <= x (dstebz: TMP1 = D(J) - E2(J-1) / TMP1 - X with the pivmin guard)
__device__ __forceinline__ int el_sturm(const double* __restrict__ d, const double* __restrict__ e2, int n, double x, double pivmin) {
  double q = d[0] - x;
  if (fabs(q) < pivmin) q = -pivmin;
  int c = q <= 0.0 ? 1 : 0;
  for (int j = 1; j < n; ++j) {
    q = d[j] - e2[j - 1] / q - x;
    if (fabs(q) < pivmin) q = -pivmin;
    c += q <= 0.0 ? 1 : 0;
  }
  return c;
}

// scal[0] = ||T||_1 (for the inverse iteration); lam ascending by construction (eigenvalue i = the one with Sturm index i)
__global__ __launch_bounds__(EL_NT) void el_bisect_kernel(const double* __restrict__ dvec, const double* __restrict__ evec, int n,
                                                          double* __restrict__ lam, double* __restrict__ scal) {
  extern __shared__ double el_sm[];
  double* d = el_sm;            // n
  double* e2 = el_sm + n;       // n (last unused)
  double* red = e2 + n;         // 4 * 4
  const int tid = (int)threadIdx.x;
  double gl = 1.7976931348623157e308, gu = -1.7976931348623157e308, e2max = 0.0, onenrm = 0.0;
  for (int j = tid; j < n; j += EL_NT) {
    const double dj = dvec[j];
    const double el = j > 0 ? fabs(evec[j - 1]) : 0.0, er = j < n - 1 ? fabs(evec[j]) : 0.0;
    d[j] = dj;
    e2[j] = er * er;
    gl = fmin(gl, dj - el - er);
    gu = fmax(gu, dj + el + er);
    e2max = fmax(e2max, er * er);
    onenrm = fmax(onenrm, fabs(dj) + el + er);
  }
  // min / max over the workgroup (order does not matter for min / max)
  for (int o = 32; o > 0; o >>= 1) {
    gl = fmin(gl, __shfl_xor(gl, o, 64)); gu = fmax(gu, __shfl_xor(gu, o, 64));
    e2max = fmax(e2max, __shfl_xor(e2max, o, 64)); onenrm = fmax(onenrm, __shfl_xor(onenrm, o, 64));
  }
  if ((tid & 63) == 0) { const int w = tid >> 6; red[4 * w] = gl; red[4 * w + 1] = gu; red[4 * w + 2] = e2max; red[4 * w + 3] = onenrm; }
  __syncthreads();
  gl = fmin(fmin(red[0], red[4]), fmin(red[8], red[12]));
  gu = fmax(fmax(red[1], red[5]), fmax(red[9], red[13]));
  e2max = fmax(fmax(red[2], red[6]), fmax(red[10], red[14]));
  onenrm = fmax(fmax(red[3], red[7]), fmax(red[11], red[15]));
  const double pivmin = EL_SAFMIN * fmax(1.0, e2max);
  const double tnorm = fmax(fabs(gl), fabs(gu));
  gl = gl - 2.1 * tnorm * (2.0 * EL_EPS) * n - 4.2 * pivmin;         // dstebz widens the Gershgorin interval (ulp = 2 eps)
  gu = gu + 2.1 * tnorm * (2.0 * EL_EPS) * n + 4.2 * pivmin;
  const int i = (int)blockIdx.x * EL_NT + tid;
  if (i == 0) scal[0] = onenrm;
  if (i >= n) return;
  double lo = gl, hi = gu;
  for (int it = 0; it < 200; ++it) {
    // dlaebz: converged when the interval is below max(abstol = ulp ||T||, pivmin, 2 ulp max(|lo|, |hi|))
    if (hi - lo <= fmax(2.0 * EL_EPS * tnorm, fmax(pivmin, 4.0 * EL_EPS * fmax(fabs(lo), fabs(hi))))) break;
    const double mid = 0.5 * (lo + hi);
    if (!(mid > lo && mid < hi)) break;
    if (el_sturm(d, e2, n, mid, pivmin) >= i + 1) hi = mid; else lo = mid;
  }
  lam[i] = 0.5 * (lo + hi);
}

// dstein: eigenvalues closer than 10 eps |lambda| are moved apart by that much, in ascending order (one thread: n steps)
__global__ void el_perturb_kernel(const double* __restrict__ lam, double* __restrict__ lamp, int n) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double xjm = 0.0;
  for (int j = 0; j < n; ++j) {
    double xj = lam[j];
    if (j > 0) {
      const double pertol = 10.0 * fabs(EL_EPS * xj);
      if (xj - xjm < pertol) xj = xjm + pertol;
    }
    lamp[j] = xj;
    xjm = xj;
  }
}

// ---- 3. inverse iteration ----------------------------------------------------------------------------------------------
__device__ __forceinline__ double el_rand(unsigned i, unsigned j) {     // uniform in (-1, 1), a hash of (vector, component)
  unsigned long long z = ((unsigned long long)i << 32 | j) + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return ((double)(z >> 11) + 0.5) * (2.0 / 9007199254740992.0) - 1.0;
}

// Thread i: eigenvector i of T.  work: five arrays [n][NTH] (a, b, c, dd, in), Z: [n][NTH] (component j of vector i at
// Z[j * NTH + i]) -- every access of a wavefront is one contiguous segment.
__global__ __launch_bounds__(EL_NT) void el_invit_kernel(const double* __restrict__ dvec, const double* __restrict__ evec, int n,
                                                         const double* __restrict__ lamp, const double* __restrict__ scal,
                                                         double* __restrict__ work, double* __restrict__ Z, long long NTH) {
  const int i = (int)blockIdx.x * EL_NT + (int)threadIdx.x;
  if (i >= n) return;
  const size_t S = (size_t)n * (size_t)NTH;
  double* a = work + i;
  double* b = work + S + i;
  double* c = work + 2 * S + i;
  double* dd = work + 3 * S + i;
  double* in = work + 4 * S + i;
  double* x = Z + i;
  const double sigma = lamp[i];
  const double onenrm = scal[0] > 0.0 ? scal[0] : 1.0;      // the zero matrix: any scale gives a start vector that survives
  // ---- dlagtf: T - sigma I = P L U
  double tolmax;
  {
    double ak = dvec[0] - sigma, bk = n > 1 ? evec[0] : 0.0;
    double scale1 = fabs(ak) + fabs(bk);
    tolmax = 0.0;
    for (int k = 0; k < n - 1; ++k) {
      const double ck = evec[k];
      double ak1 = dvec[k + 1] - sigma;
      double bk1 = k < n - 2 ? evec[k + 1] : 0.0;
      double scale2 = fabs(ck) + fabs(ak1);
      if (k < n - 2) scale2 += fabs(bk1);
      const double piv1 = ak == 0.0 ? 0.0 : fabs(ak) / scale1;
      double a_st, b_st, c_st, d_st = 0.0, in_st = 0.0;
      if (ck == 0.0) {
        a_st = ak; b_st = bk; c_st = 0.0;
        scale1 = scale2;
      } else {
        const double piv2 = fabs(ck) / scale2;
        if (piv2 <= piv1) {
          scale1 = scale2;
          c_st = ck / ak;
          a_st = ak; b_st = bk;
          ak1 = ak1 - c_st * bk;
        } else {
          in_st = 1.0;
          const double mult = ak / ck;
          a_st = ck;
          const double temp = ak1;
          ak1 = bk - mult * temp;
          if (k < n - 2) { d_st = bk1; bk1 = -mult * d_st; }
          b_st = temp;
          c_st = mult;
        }
      }
      a[(size_t)k * NTH] = a_st; b[(size_t)k * NTH] = b_st; c[(size_t)k * NTH] = c_st; dd[(size_t)k * NTH] = d_st; in[(size_t)k * NTH] = in_st;
      tolmax = fmax(tolmax, fmax(fabs(a_st), fmax(fabs(b_st), fabs(d_st))));
      ak = ak1; bk = bk1;
    }
    a[(size_t)(n - 1) * NTH] = ak;
    tolmax = fmax(tolmax, fabs(ak));
  }
  const double a_last = a[(size_t)(n - 1) * NTH];
  double tol = tolmax * EL_EPS;                       // dlagts: tol = eps * max |U|
  if (tol == 0.0) tol = EL_EPS;
  const double bignum = 1.0 / EL_SAFMIN;
  // ---- dstein: random start, scaled inverse iteration until the growth test passes, then EXTRA more
  double xmax = 0.0;
  for (int j = 0; j < n; ++j) { const double r = el_rand((unsigned)i, (unsigned)j); x[(size_t)j * NTH] = r; xmax = fmax(xmax, fabs(r)); }
  const double dtpcrt = sqrt(0.1 / n);
  constexpr int MAXITS = 5, EXTRA = 1;
  int nrmchk = 0;
  for (int its = 0; its < MAXITS; ++its) {
    const double scl = n * onenrm * fmax(EL_EPS, fabs(a_last)) / (xmax > 0.0 ? xmax : 1.0);
    // forward: L^-1 P (scl x), fused with the scaling
    double yprev = x[0] * scl;
    for (int k = 1; k < n; ++k) {
      double yk = x[(size_t)k * NTH] * scl;
      const double ck = c[(size_t)(k - 1) * NTH];
      if (in[(size_t)(k - 1) * NTH] == 0.0) {
        yk -= ck * yprev;
      } else {
        const double temp = yprev;
        yprev = yk;
        yk = temp - ck * yk;
      }
      x[(size_t)(k - 1) * NTH] = yprev;
      yprev = yk;
    }
    x[(size_t)(n - 1) * NTH] = yprev;
    // backward: U^-1 with dlagts' perturbation of unusably small pivots
    double y1 = 0.0, y2 = 0.0;
    xmax = 0.0;
    for (int k = n - 1; k >= 0; --k) {
      double temp = x[(size_t)k * NTH];
      if (k <= n - 2) temp -= b[(size_t)k * NTH] * y1;
      if (k <= n - 3) temp -= dd[(size_t)k * NTH] * y2;
      double ak = a[(size_t)k * NTH];
      double pert = copysign(tol, ak);
      for (;;) {
        const double absak = fabs(ak);
        if (absak < 1.0) {
          if (absak < EL_SAFMIN) {
            if (absak == 0.0 || fabs(temp) * EL_SAFMIN > absak) { ak += pert; pert *= 2.0; continue; }
            temp *= bignum; ak *= bignum;
          } else if (fabs(temp) > absak * bignum) { ak += pert; pert *= 2.0; continue; }
        }
        break;
      }
      const double yk = temp / ak;
      x[(size_t)k * NTH] = yk;
      xmax = fmax(xmax, fabs(yk));
      y2 = y1; y1 = yk;
    }
    if (!(xmax >= dtpcrt)) continue;
    if (++nrmchk < EXTRA + 1) continue;
    break;
  }
  // 2-norm 1 (scaled sum of squares: the iterate may be huge)
  double ssq = 0.0;
  const double inv = xmax > 0.0 ? 1.0 / xmax : 0.0;
  for (int j = 0; j < n; ++j) { const double v = x[(size_t)j * NTH] * inv; ssq += v * v; }
  const double s = ssq > 0.0 ? inv / sqrt(ssq) : 0.0;
  for (int j = 0; j < n; ++j) x[(size_t)j * NTH] *= s;
}

// ---- 4. Cholesky-QR helpers -------------------------------------------------------------------------------------------
// G += I on the padding rows (i >= n), G += shift * I elsewhere; err[0] = max |G - I| over the n x n part before the change
__global__ __launch_bounds__(EL_NT) void el_gram_fix_kernel(double* __restrict__ G, int K, int n, double shift, double* __restrict__ err) {
  __shared__ double red[4];
  const int row = (int)blockIdx.x;
  double m = 0.0;
  if (row < n)
    for (int c = (int)threadIdx.x; c < n; c += EL_NT) m = fmax(m, fabs(G[(size_t)row * K + c] - (c == row ? 1.0 : 0.0)));
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    err[1 + row] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    if (row >= n) G[(size_t)row * K + row] = 1.0; else if (shift != 0.0) G[(size_t)row * K + row] += shift;
  }
}
__global__ __launch_bounds__(EL_NT) void el_max_kernel(double* __restrict__ err, int K, const double* __restrict__ dd, int n) {
  __shared__ double red[4], redm[4];
  double m = 0.0, dmin = 1.7976931348623157e308;
  for (int i = (int)threadIdx.x; i < K; i += EL_NT) m = fmax(m, err[1 + i]);
  if (dd) for (int i = (int)threadIdx.x; i < n; i += EL_NT) { const double d = dd[i]; dmin = (d == d) ? fmin(dmin, d) : -1.0; }
  for (int o = 32; o > 0; o >>= 1) { m = fmax(m, __shfl_xor(m, o, 64)); dmin = fmin(dmin, __shfl_xor(dmin, o, 64)); }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = m; redm[threadIdx.x >> 6] = dmin; }
  __syncthreads();
  if (threadIdx.x == 0) {
    err[0] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    err[K + 1] = fmin(fmin(redm[0], redm[1]), fmin(redm[2], redm[3]));     // smallest pivot of the factorisation that just ran
  }
}
// rows of M scaled by 1 / sqrt(d_i) (i < n)
__global__ __launch_bounds__(EL_NT) void el_row_scale_kernel(double* __restrict__ M, int K, int n, const double* __restrict__ dd) {
  const int row = (int)blockIdx.x;
  if (row >= n) return;
  const double s = 1.0 / sqrt(dd[row]);
  for (int c = (int)threadIdx.x; c < K; c += EL_NT) M[(size_t)row * K + c] *= s;
}

// ---- 5. back-transformation ---------------------------------------------------------------------------------------------
// M: row r = eigenvector r of T (K x K row-major).  V[:, r] = H_0 H_1 ... H_{n-2} M[r, :]^T, written column-major (ld n).
template <int NC>
__global__ __launch_bounds__(EL_NT) void el_backtransform_kernel(const double* __restrict__ M, int K, int n, const double* __restrict__ H,
                                                                 long long ldh, const double* __restrict__ tau, double* __restrict__ V) {
  extern __shared__ double el_sm[];
  double* z = el_sm;                        // NC x n
  double* red = el_sm + (size_t)NC * n;     // 4 x NC
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = (int)blockIdx.x * NC;
  for (int c = 0; c < NC; ++c) {
    const int r = r0 + c;
    for (int j = tid; j < n; j += EL_NT) z[(size_t)c * n + j] = r < n ? M[(size_t)r * K + j] : 0.0;
  }
  __syncthreads();
  for (int k = n - 2; k >= 0; --k) {
    const double tk = tau[k];
    if (tk == 0.0) continue;
    const int t = n - k - 1;
    const double* v = H + (size_t)k * ldh + (k + 1);
    double s[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) s[c] = 0.0;
    for (int i = tid; i < t; i += EL_NT) {
      const double vi = v[i];
#pragma unroll
      for (int c = 0; c < NC; ++c) s[c] += vi * z[(size_t)c * n + k + 1 + i];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      s[c] = wave_sum(s[c]);
      if (lane == 0) red[wave * NC + c] = s[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NC; ++c) s[c] = tk * ((red[c] + red[NC + c]) + (red[2 * NC + c] + red[3 * NC + c]));
    for (int i = tid; i < t; i += EL_NT) {
      const double vi = v[i];
#pragma unroll
      for (int c = 0; c < NC; ++c) z[(size_t)c * n + k + 1 + i] -= s[c] * vi;
    }
    __syncthreads();
  }
  for (int c = 0; c < NC; ++c) {
    const int r = r0 + c;
    if (r >= n) break;
    for (int j = tid; j < n; j += EL_NT) V[(size_t)r * n + j] = z[(size_t)c * n + j];
  }
}

template <typename KernT>
int el_allow_lds(KernT kern, size_t bytes) {
  // per kernel and process-wide: always the hardware maximum (a later, smaller call must not lower it under a concurrent larger one)
  if (bytes > 48 * 1024) CUADMM_HIP_TRY(allow_max_dynamic_lds(reinterpret_cast<const void*>(kern)));
  return CUADMM_OK;
}

// ---- projection of one block through the eigendecomposition -------------------------------------------------------------------
// svec (column-major upper triangle, off-diagonal scaled by sqrt 2) -> dense column-major symmetric
__global__ __launch_bounds__(EL_NT) void el_unpack_svec_kernel(const double* __restrict__ sv, int n, double* __restrict__ A) {
  const int c = (int)blockIdx.x;
  const double* col = sv + (long long)c * (c + 1) / 2;
  for (int r = (int)threadIdx.x; r <= c; r += EL_NT) {
    const double v = r == c ? col[r] : col[r] * 0.70710678118654752440;
    A[(size_t)c * n + r] = v;
    A[(size_t)r * n + c] = v;
  }
}
// out = svec(sum_{k >= k0} max(W_k, 0) v_k v_k^T): one workgroup per column c of the upper triangle, the eigenvalues ascending
__global__ __launch_bounds__(EL_NT) void el_rebuild_kernel(const double* __restrict__ V, const double* __restrict__ W, int n, int k0,
                                                           double* __restrict__ sv) {
  extern __shared__ double el_sm[];
  double* vc = el_sm;                    // lambda_k^+ V[c, k], k = k0 .. n-1
  const int c = (int)blockIdx.x;
  int kpos = k0;
  // ascending eigenvalues: the positive ones are a suffix; its start is found by every thread (n reads of a cached array at most)
  while (kpos < n && !(W[kpos] > 0.0)) ++kpos;
  for (int k = kpos + (int)threadIdx.x; k < n; k += EL_NT) vc[k - kpos] = W[k] * V[(size_t)k * n + c];
  __syncthreads();
  double* col = sv + (long long)c * (c + 1) / 2;
  for (int r = (int)threadIdx.x; r <= c; r += EL_NT) {
    double acc0 = 0.0, acc1 = 0.0;
    int k = kpos;
    for (; k + 1 < n; k += 2) { acc0 += V[(size_t)k * n + r] * vc[k - kpos]; acc1 += V[(size_t)(k + 1) * n + r] * vc[k + 1 - kpos]; }
    if (k < n) acc0 += V[(size_t)k * n + r] * vc[k - kpos];
    const double v = acc0 + acc1;
    col[r] = r == c ? v : v * 1.41421356237309504880;
  }
}
__global__ void el_flag_fail_kernel(const int* __restrict__ info, int* __restrict__ fail) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && *info != 0) atomicAdd(fail, 1);
}

}  // namespace

int eig_large(double* mat, double* W, int* info, int n, hipStream_t st, EigLargeWs* ws) {
  if (n < 2 || n > kEigLargeMax) { set_error("eig_large: n = %d outside [2, %d]", n, kEigLargeMax); return CUADMM_ERR_INVALID; }
  const int K = (n + 63) / 64 * 64;
  const size_t KK = (size_t)K * K;
  EigLargeWs local;
  if (!ws) ws = &local;
  int rc = ws->ensure(n);
  if (rc) return rc;
  double *H = ws->H, *tau = ws->tau, *dvec = ws->dvec, *evec = ws->evec, *P = ws->P, *lamp = ws->lamp, *scal = ws->scal, *work = ws->work,
         *Z = ws->Z, *M = ws->M, *Mt = ws->Mt, *G = ws->G, *Winv = ws->Winv, *Wtmp = ws->Wtmp, *dd = ws->dd, *Yp = ws->Yp, *err = ws->err;
  int* dflag = ws->dflag;
  const long long ld = n;

  // ---- 1. tridiagonalisation (in place: `mat` is destroyed, the eigenvectors replace it at the end)
  {
    const size_t lds_a = sizeof(double) * ((size_t)n + 8), lds_b = sizeof(double) * (2 * (size_t)n + 8);
    if ((rc = el_allow_lds(el_house_symv_kernel, lds_a)) || (rc = el_allow_lds(el_rank2_kernel, lds_b))) return rc;
    for (int k = 0; k < n - 1; ++k) {
      const int t = n - k - 1, grid = (t + EL_CPW - 1) / EL_CPW;
      hipLaunchKernelGGL(el_house_symv_kernel, dim3(grid), dim3(EL_NT), sizeof(double) * ((size_t)t + 8), st, mat, ld, n, k, H, tau, dvec, evec, P);
      hipLaunchKernelGGL(el_rank2_kernel, dim3(grid), dim3(EL_NT), sizeof(double) * (2 * (size_t)t + 8), st, mat, ld, n, k, H, tau, P);
    }
    hipLaunchKernelGGL(el_last_diag_kernel, dim3(1), dim3(64), 0, st, mat, ld, n, dvec);
    CUADMM_HIP_TRY(hipGetLastError());
  }
  // ---- 2. / 3. eigenvalues by bisection, eigenvectors by inverse iteration
  {
    const size_t lds = sizeof(double) * (2 * (size_t)n + 16);
    if ((rc = el_allow_lds(el_bisect_kernel, lds))) return rc;
    const int grid = (n + EL_NT - 1) / EL_NT;
    hipLaunchKernelGGL(el_bisect_kernel, dim3(grid), dim3(EL_NT), lds, st, dvec, evec, n, W, scal);
    hipLaunchKernelGGL(el_perturb_kernel, dim3(1), dim3(64), 0, st, W, lamp, n);
    CUADMM_HIP_TRY(hipMemsetAsync(Z, 0, sizeof(double) * KK, st));
    hipLaunchKernelGGL(el_invit_kernel, dim3(grid), dim3(EL_NT), 0, st, dvec, evec, n, lamp, scal, work, Z, (long long)K);
    CUADMM_HIP_TRY(hipGetLastError());
  }
  // ---- 4. Cholesky-QR passes: M (row r = vector r) and Mt = M^T = Z
  if ((rc = ts_transpose(Z, M, K, st))) return rc;
  double* cur_t = Z;                       // M^T of the current M
  int converged = 0;
  double shift = 0.0;
  for (int pass = 0; pass < 6 && !converged; ++pass) {
    if ((rc = ts_gemm(K, K, K, 1.0, M, K, 0, cur_t, K, 0, G, K, 0, 1, st))) return rc;
    hipLaunchKernelGGL(el_gram_fix_kernel, dim3(K), dim3(EL_NT), 0, st, G, K, n, shift, err);
    CUADMM_HIP_TRY(hipMemsetAsync(dflag, 0, sizeof(int), st));
    // the convergence test needs max |G - I| of THIS Gram matrix: read it before deciding to factor
    hipLaunchKernelGGL(el_max_kernel, dim3(1), dim3(EL_NT), 0, st, err, K, (const double*)nullptr, n);
    double h_err = 0.0;
    if ((rc = staged_d2h(&h_err, err, sizeof(double), st))) return rc;
    if (h_err != h_err) { set_error("eig_large: non-finite entries (input or inverse iteration)"); return CUADMM_ERR_EIG; }
    if (shift == 0.0 && h_err <= 1e-13) { converged = 1; break; }
    // G = L D L^T, M <- D^-1/2 L^-1 M
    if ((rc = ts_ldlt_factor(G, K, dd, Yp, dflag, st))) return rc;
    hipLaunchKernelGGL(el_max_kernel, dim3(1), dim3(EL_NT), 0, st, err, K, (const double*)dd, n);
    double h_dmin = 0.0;
    int h_flag = 0;
    if ((rc = staged_d2h(&h_dmin, err + K + 1, sizeof(double), st)) || (rc = staged_d2h(&h_flag, dflag, sizeof(int), st))) return rc;
    if (h_flag != 0 || !(h_dmin > 0.0)) {
      // numerically dependent vectors: shifted Cholesky-QR (the pass after a shifted one is unshifted again)
      shift = shift == 0.0 ? 11.0 * EL_EPS * (double)n * (double)n : shift * 100.0;
      if (shift > 1e-2) break;
      continue;                            // G is recomputed from M with the shift on its diagonal
    }
    shift = 0.0;
    if ((rc = ts_unit_lower_inverse(G, Winv, Wtmp, K, st))) return rc;
    if ((rc = ts_gemm(K, K, K, 1.0, Winv, K, 0, M, K, 0, Mt, K, 0, 1, st))) return rc;     // Mt used as the output buffer
    hipLaunchKernelGGL(el_row_scale_kernel, dim3(K), dim3(EL_NT), 0, st, Mt, K, n, dd);
    std::swap(M, Mt);
    if ((rc = ts_transpose(M, Mt, K, st))) return rc;
    cur_t = Mt;
    CUADMM_HIP_TRY(hipGetLastError());
  }
  // ---- 5. back-transformation into `mat`
  {
    const size_t per_col = sizeof(double) * (size_t)n;
    int nc = (int)std::min<size_t>(8, (kMaxLdsBytes - 512) / per_col);
    nc = nc >= 8 ? 8 : (nc >= 4 ? 4 : (nc >= 2 ? 2 : 1));
    const size_t lds = per_col * nc + sizeof(double) * 4 * nc;
    const int grid = (n + nc - 1) / nc;
    switch (nc) {
      case 8: if ((rc = el_allow_lds(el_backtransform_kernel<8>, lds))) return rc;
              hipLaunchKernelGGL(el_backtransform_kernel<8>, dim3(grid), dim3(EL_NT), lds, st, M, K, n, H, ld, tau, mat); break;
      case 4: if ((rc = el_allow_lds(el_backtransform_kernel<4>, lds))) return rc;
              hipLaunchKernelGGL(el_backtransform_kernel<4>, dim3(grid), dim3(EL_NT), lds, st, M, K, n, H, ld, tau, mat); break;
      case 2: if ((rc = el_allow_lds(el_backtransform_kernel<2>, lds))) return rc;
              hipLaunchKernelGGL(el_backtransform_kernel<2>, dim3(grid), dim3(EL_NT), lds, st, M, K, n, H, ld, tau, mat); break;
      default: if ((rc = el_allow_lds(el_backtransform_kernel<1>, lds))) return rc;
               hipLaunchKernelGGL(el_backtransform_kernel<1>, dim3(grid), dim3(EL_NT), lds, st, M, K, n, H, ld, tau, mat); break;
    }
    CUADMM_HIP_TRY(hipGetLastError());
  }
  if (info) {
    const int v = converged ? 0 : 1;
    if ((rc = staged_h2d(info, &v, sizeof(int), st))) return rc;
  }
  CUADMM_HIP_TRY(hipStreamSynchronize(st));
  return CUADMM_OK;
}

int EigLargeWs::ensure(int n) {
  if (n <= cap) return CUADMM_OK;
  release();
  const size_t K = ((size_t)n + 63) / 64 * 64, KK = K * K;
  auto get = [&](auto** out, size_t count) -> int {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, sizeof(**out) * std::max<size_t>(count, 1));
    if (e != hipSuccess) { set_error("eig_large: workspace allocation failed (%s)", hipGetErrorString(e)); return CUADMM_ERR_NO_DEVICE; }
    bufs.push_back(q);
    *out = static_cast<std::remove_reference_t<decltype(**out)>*>(q);
    return CUADMM_OK;
  };
  int rc;
  if ((rc = get(&H, (size_t)n * n)) || (rc = get(&tau, n)) || (rc = get(&dvec, n)) || (rc = get(&evec, n)) || (rc = get(&P, n)) ||
      (rc = get(&lamp, n)) || (rc = get(&scal, 4)) || (rc = get(&work, 5 * (size_t)n * K)) || (rc = get(&Z, KK)) || (rc = get(&M, KK)) ||
      (rc = get(&Mt, KK)) || (rc = get(&G, KK)) || (rc = get(&Winv, KK)) || (rc = get(&Wtmp, KK)) || (rc = get(&dd, K)) ||
      (rc = get(&Yp, K * 64)) || (rc = get(&err, K + 2)) || (rc = get(&dflag, 1)) || (rc = get(&dense, (size_t)n * n)) || (rc = get(&Wd, n))) {
    release();
    return rc;
  }
  cap = n;
  return CUADMM_OK;
}

void EigLargeWs::release() {
  for (void* q : bufs) { hipError_t e = hipFree(q); (void)e; }
  bufs.clear();
  cap = 0;
}

int eig_large_project(const double* svec_in, double* svec_out, int n, int eig_rank, int* fail, hipStream_t st, EigLargeWs* ws) {
  if (!ws) { set_error("eig_large_project: no workspace"); return CUADMM_ERR_INVALID; }
  int rc = ws->ensure(n);
  if (rc) return rc;
  hipLaunchKernelGGL(el_unpack_svec_kernel, dim3(n), dim3(EL_NT), 0, st, svec_in, n, ws->dense);
  CUADMM_HIP_TRY(hipGetLastError());
  if ((rc = eig_large(ws->dense, ws->Wd, ws->dflag, n, st, ws))) return rc;
  if (fail) hipLaunchKernelGGL(el_flag_fail_kernel, dim3(1), dim3(64), 0, st, ws->dflag, fail);
  const int k0 = eig_rank > 0 ? std::max(0, n - eig_rank) : 0;
  const size_t lds = sizeof(double) * (size_t)n;
  if ((rc = el_allow_lds(el_rebuild_kernel, lds))) return rc;
  hipLaunchKernelGGL(el_rebuild_kernel, dim3(n), dim3(EL_NT), lds, st, ws->dense, ws->Wd, n, k0, svec_out);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

}  // namespace cuadmm
