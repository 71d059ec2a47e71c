// Device-side symmetric eigensolver + PSD projection shared by the three projection kernels
// (sub-wave groups for n<=32, one workgroup per block with the matrix in LDS for mid sizes,
// one workgroup per block with the matrix in an HBM workspace for the rest).
//
// Replaces, per block, the reference pipeline solver.cu:534-647:
//   vector_to_matrices -> cusolverDnXsyevd / DsyevjBatched -> max(W,0) -> V*diag(W) -> DGEMM(N,T)
//   -> matrices_to_vector
// Algorithm (scalar CPU twin: oracle/eigproj_twin.c): Householder tridiagonalisation with the
// reflectors kept in the lower triangle, Q formed in place by backward accumulation, implicit
// QL with Wilkinson shift accumulating rotations into Q's columns; then P = Z diag(l+) Z^T.
//
// Layout: M is n x n ROW-major with leading dimension ld (odd, so that both "lane r reads
// M[r][k]" and "lane c reads M[r][c]" are LDS-bank-conflict free for 8-byte accesses).
// Thread `rank` of a group owns rows/columns rank, rank+GS, ...
#pragma once
#include <hip/hip_runtime.h>

namespace cuadmm {

constexpr double kSqrt2 = 0x1.6a09e667f3bccp+0;     // reference SQRT2    (include/cuadmm/kernels.h:180)
constexpr double kSqrt2Inv = 0x1.6a09e667f3bcdp-1;  // reference SQRT2INV (include/cuadmm/kernels.h:181)
constexpr int kQlMaxSweepsPerEig = 60;

// The eigen-solver below is also compiled for the host with a one-thread group policy
// (tools/host_check_eig.cpp) so that its control flow can be debugged without a GPU.
#define CUADMM_HD __host__ __device__

CUADMM_HD __forceinline__ void wave_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#endif
}

CUADMM_HD __forceinline__ int lane_id() {
#if defined(__HIP_DEVICE_COMPILE__)
  return (int)(threadIdx.x & 63u);
#else
  return 0;
#endif
}

// sqrt(h) and 1/sqrt(h) together from one v_rsq_f64 + two Goldschmidt steps (h > 0, normal range).
// ~1-2 ulp; replaces an IEEE sqrt followed by an IEEE divide (about 45 dependent instructions) on the
// serial critical path of the QL recurrence.
CUADMM_HD __forceinline__ void fast_sqrt_rsqrt(double h, double& root, double& inv_root) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double y = __builtin_amdgcn_rsq(h);
#else
  const double y = (double)(1.0f / sqrtf((float)h));   // host twin: a low-precision seed, like v_rsq_f64
#endif
  double g = h * y, hh = 0.5 * y;
  double r = fma(-hh, g, 0.5);
  g = fma(g, r, g); hh = fma(hh, r, hh);
  r = fma(-hh, g, 0.5);
  g = fma(g, r, g); hh = fma(hh, r, hh);
  root = g;
  inv_root = hh + hh;
}

// 1/x from v_rcp_f64 + two Newton steps (x finite, non-zero, normal range); ~1 ulp
CUADMM_HD __forceinline__ double fast_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(x);
#else
  double y = (double)(1.0f / (float)x);
#endif
  double e = fma(-x, y, 1.0);
  y = fma(y, e, y);
  e = fma(-x, y, 1.0);
  return fma(y, e, y);
}

// svec slot e (0-based, within a block) -> (col i, row j), j<=i, e = i(i+1)/2 + j
__device__ __forceinline__ void tri_decode(int e, int& i, int& j) {
  int t = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
  while (t * (t + 1) / 2 > e) --t;
  while ((t + 1) * (t + 2) / 2 <= e) ++t;
  i = t;
  j = e - t * (t + 1) / 2;
}

// ---------------------------------------------------------------------------------------
// Group policies
// ---------------------------------------------------------------------------------------
// LPB lanes of a wavefront form a group; 64/LPB groups (= blocks) per wavefront.
template <int LPB>
struct SubGroup {
  static constexpr int kSize = LPB;
  static constexpr bool kMultiWave = false;
  __device__ static int rank() { return (int)(threadIdx.x & (unsigned)(LPB - 1)); }
  __device__ static int group_in_wave() { return lane_id() / LPB; }
  __device__ static double sum(double x, double* /*scratch*/) {
#pragma unroll
    for (int o = LPB >> 1; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
  }
  __device__ static void sync() { wave_fence(); }
  // first index idx in [l, n) with pred(idx); every lane evaluates pred at idx = rank
  template <class Pred>
  __device__ static int first_true(int l, int n, Pred pred) {
    int idx = rank();
    bool t = (idx >= l) && (idx < n) && pred(idx);
    unsigned long long mask = __ballot(t);
    unsigned long long bits = mask >> (group_in_wave() * LPB);
    if (LPB < 64) bits &= ((1ull << LPB) - 1ull);
    return bits ? (int)__builtin_ctzll(bits) : n;
  }
};

// A whole workgroup of NT threads works on one block.
template <int NT>
struct WgGroup {
  static constexpr int kSize = NT;
  static constexpr bool kMultiWave = true;
  __device__ static int rank() { return (int)threadIdx.x; }
  __device__ static double sum(double x, double* scratch) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    constexpr int NW = NT / 64;
    if (lane_id() == 0) scratch[threadIdx.x >> 6] = x;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += scratch[w];
    __syncthreads();
    return t;
  }
  __device__ static void sync() { __syncthreads(); }
  // wave-private search (each wave runs the QL recurrence redundantly on its own copy of d/e)
  template <class Pred>
  __device__ static int first_true(int l, int n, Pred pred) {
    for (int base = l; base < n; base += 64) {
      int idx = base + lane_id();
      bool t = (idx < n) && pred(idx);
      unsigned long long mask = __ballot(t);
      if (mask) return base + (int)__builtin_ctzll(mask);
    }
    return n;
  }
};

// ---------------------------------------------------------------------------------------
// Symmetric eigendecomposition, in place.
//   M      n x n row-major (ld), full symmetric on entry; eigenvectors Z[r][k] on exit
//   dsh/esh/tau/vv/ww  group-shared vectors of n doubles
//   dq/eq  vectors the QL recurrence runs on: == dsh/esh for SubGroup, wave-private for WgGroup
//   scratch: >= NT/64 doubles (WgGroup reductions)
// Returns 0, or 1 when QL exceeded its sweep cap (eigenvalues in dq either way).
// ---------------------------------------------------------------------------------------
template <class Gp>
CUADMM_HD __forceinline__ int sym_eig_inplace(double* __restrict__ M, const int ld, const int n, double* dsh, double* esh,
                               double* tau, double* vv, double* ww, double* dq, double* eq, double* scratch) {
  constexpr int GS = Gp::kSize;
  const int rank = Gp::rank();

  // ---- Householder tridiagonalisation -------------------------------------------------
  for (int k = 0; k < n - 2; ++k) {
    double part = 0.0;
    for (int r = rank; r < n; r += GS)
      if (r >= k + 2) { double x = M[r * ld + k]; part += x * x; }
    const double xn2 = Gp::sum(part, scratch);
    const double alpha = M[(k + 1) * ld + k];
    double t, beta, scal = 0.0;
    if (xn2 == 0.0) {
      t = 0.0; beta = alpha;
    } else {
      beta = -copysign(sqrt(alpha * alpha + xn2), alpha);
      t = (beta - alpha) / beta;
      scal = 1.0 / (alpha - beta);
    }
    if (t != 0.0) {
      for (int r = rank; r < n; r += GS) {
        if (r >= k + 2) { double v = M[r * ld + k] * scal; M[r * ld + k] = v; vv[r] = v; }
        else if (r == k + 1) vv[r] = 1.0;
      }
    }
    if (rank == 0) { esh[k] = beta; tau[k] = t; }
    Gp::sync();
    if (t != 0.0) {
      double kpart = 0.0;
      for (int r = rank; r < n; r += GS)
        if (r >= k + 1) {
          const double* row = M + r * ld;
          double p0 = 0.0, p1 = 0.0;
          int c = k + 1;
          for (; c + 1 < n; c += 2) { p0 += row[c] * vv[c]; p1 += row[c + 1] * vv[c + 1]; }
          if (c < n) p0 += row[c] * vv[c];
          double p = (p0 + p1) * t;
          ww[r] = p;
          kpart += p * vv[r];
        }
      const double K = Gp::sum(kpart, scratch) * (-0.5 * t);
      for (int r = rank; r < n; r += GS)
        if (r >= k + 1) ww[r] += K * vv[r];
      Gp::sync();
      for (int r = rank; r < n; r += GS)
        if (r >= k + 1) {
          double* row = M + r * ld;
          const double vr = vv[r], wr = ww[r];
          for (int c = k + 1; c < n; ++c) row[c] -= vr * ww[c] + wr * vv[c];
        }
      Gp::sync();
    }
  }
  for (int r = rank; r < n; r += GS) dsh[r] = M[r * ld + r];
  if (rank == 0) {
    if (n >= 2) esh[n - 2] = M[(n - 1) * ld + (n - 2)];
    esh[n - 1] = 0.0;
  }
  Gp::sync();
  // Deflation threshold relative to the norm of the whole tridiagonal matrix (as EISPACK tql2 does
  // with its running max): rotations against the largest entries leave absolute noise ~eps*||T||
  // in every off-diagonal, so on strongly graded / rank-deficient blocks (moment matrices: one
  // eigenvalue O(1), the rest ~1e-15) a purely local relative test is never met.
  double tpart = 0.0;
  for (int r = rank; r < n; r += GS) { const double dv = dsh[r], ev = esh[r]; tpart += dv * dv + 2.0 * ev * ev; }
  const double eps_abs = sqrt(Gp::sum(tpart, scratch)) * 0x1p-53;

  // ---- form Q in place ------------------------------------------------------------------
  if (rank == 0) M[(n - 1) * ld + (n - 1)] = 1.0;
  for (int k = n - 3; k >= 0; --k) {
    const double t = tau[k];
    for (int r = rank; r < n; r += GS) {
      if (r >= k + 2) {
        vv[r] = M[r * ld + k];
        M[(k + 1) * ld + r] = 0.0;
        M[r * ld + (k + 1)] = 0.0;
      } else if (r == k + 1) {
        vv[r] = 1.0;
        M[(k + 1) * ld + (k + 1)] = 1.0;
      }
    }
    Gp::sync();
    if (t != 0.0) {
      for (int c = rank; c < n; c += GS)
        if (c >= k + 1) {
          double s0 = 0.0, s1 = 0.0;
          int r = k + 1;
          for (; r + 1 < n; r += 2) { s0 += vv[r] * M[r * ld + c]; s1 += vv[r + 1] * M[(r + 1) * ld + c]; }
          if (r < n) s0 += vv[r] * M[r * ld + c];
          const double s = (s0 + s1) * t;
          for (r = k + 1; r < n; ++r) M[r * ld + c] -= vv[r] * s;
        }
    }
    Gp::sync();
  }
  if (n >= 2) {
    for (int c = rank; c < n; c += GS) {
      if (c == 0) M[0] = 1.0;
      else { M[c] = 0.0; M[c * ld] = 0.0; }
    }
  }
  Gp::sync();

  // ---- implicit QL ----------------------------------------------------------------------
  if (Gp::kMultiWave && dq != dsh) {
    for (int idx = lane_id(); idx < n; idx += 64) { dq[idx] = dsh[idx]; eq[idx] = esh[idx]; }
    wave_fence();
  }
  const bool writer = Gp::kMultiWave ? (lane_id() == 0) : (rank == 0);
  int l = 0, m = 0, sweeps = 0, fail = 0;
  bool done = (n <= 1);
  while (!done) {
    // find the first negligible off-diagonal at or after l; deflate converged eigenvalues
    for (;;) {
      m = Gp::first_true(l, n, [&](int idx) {
        if (idx >= n - 1) return true;
        const double ae = fabs(eq[idx]);
        const double dd = fabs(dq[idx]) + fabs(dq[idx + 1]);
        return ae <= eps_abs || ae + dd == dd;
      });
      if (m >= n) m = n - 1;
      if (m > l) break;
      ++l; sweeps = 0;
      if (l >= n) { done = true; break; }
    }
    if (done) break;
    if (sweeps++ >= kQlMaxSweepsPerEig) { fail = 1; break; }
#ifdef CUADMM_QL_CHECKS
    if (l < 0 || m <= l || m >= n) { fail = 1000 + m * 10000 + l * 10; break; }
#endif
    // Wilkinson shift
    const double dl = dq[l], el = eq[l];
    double g = (dq[l + 1] - dl) * fast_rcp(el + el);
    double r0, r0i;
    fast_sqrt_rsqrt(fma(g, g, 1.0), r0, r0i);
    g = dq[m] - dl + el * fast_rcp(g + copysign(r0, g));
    double s = 1.0, c = 1.0, p = 0.0;
    // one sweep: rotations (i, i+1) for i = m-1 .. l.  Slot i writes d[i+1], e[i+1] and slot i-1 reads
    // d[i-1], e[i-1], d[i]: no hazard between slots, so the only fence is at the end of the sweep.
    // Z is streamed: the current column i+1 of the thread's first row is carried in a register.
    double e_c = eq[m - 1], d_c = dq[m - 1], d1_c = dq[m];
    const bool own = rank < n;
    double* zrow = M + (size_t)(own ? rank : 0) * ld;   // non-owners point at a valid row: the compiler may speculate the loads
    double carry = own ? zrow[m] : 0.0;
    double z_c = own ? zrow[m - 1] : 0.0;
    int i = m - 1;
    bool broke = false;
    for (; i >= l; --i) {
      double e_n = 0.0, d_n = 0.0, z_n = 0.0;
      if (i > l) { e_n = eq[i - 1]; d_n = dq[i - 1]; if (own) z_n = zrow[i - 1]; }   // prefetch for slot i-1
      const double f = s * e_c, b = c * e_c;
      const double h = fma(f, f, g * g);
      if (h == 0.0) {                                   // underflow recovery of the textbook recurrence
        if (writer) { eq[i + 1] = 0.0; dq[i + 1] = d1_c - p; eq[m] = 0.0; }
        broke = true;
        break;
      }
      double rr, rinv;
      fast_sqrt_rsqrt(h, rr, rinv);
      s = f * rinv; c = g * rinv;
      g = d1_c - p;
      const double cb = c * b;
      const double r2 = fma(d_c - g, s, cb + cb);
      p = s * r2;
      if (writer) { eq[i + 1] = rr; dq[i + 1] = g + p; }
      g = fma(c, r2, -b);
      if (own) {
        zrow[i + 1] = fma(s, z_c, c * carry);
        carry = fma(c, z_c, -(s * carry));
      }
      for (int r = rank + GS; r < n; r += GS) {         // further rows of this thread (large blocks only)
        double* z = M + (size_t)r * ld + i;
        const double z0 = z[0], z1 = z[1];
        z[1] = fma(s, z0, c * z1);
        z[0] = fma(c, z0, -(s * z1));
      }
      d1_c = d_c; e_c = e_n; d_c = d_n; z_c = z_n;
    }
    // column i+1 of the first row is still in the register (i = l-1 after a full sweep)
    if (own) zrow[i + 1] = carry;
    if (!broke && writer) { dq[l] = dq[l] - p; eq[l] = g; eq[m] = 0.0; }
    wave_fence();
  }
  Gp::sync();
  return fail;
}

// P = (Z diag(max(d,0))) Z^T written in svec form (sqrt(2) on off-diagonals) to out[0 .. n(n+1)/2):
// the product structure of the reference, T = V*diag(W) (diagonal_batch.cu:11-23) then P = T*V^T
// (cublas.h:18-35), so that e.g. a 1x1 block returns max(x,0) exactly.  Only the upper triangle
// that svec stores is computed.
template <class Gp>
CUADMM_HD __forceinline__ void reconstruct_to_svec(const double* __restrict__ M, const int ld, const int n, const double* dq, double* vv,
                                    double* __restrict__ out, const int eig_rank = 0) {
  constexpr int GS = Gp::kSize;
  const int rank = Gp::rank();
  for (int k = rank; k < n; k += GS) {   // dense_scalar.cu:41-47; with a rank limit :51-57 + get_eig_rank_mask.cu:13-37
    const double lam = dq[k];
    double lp = lam > 0.0 ? lam : 0.0;
    if (eig_rank > 0) {                  // keep only the eig_rank largest eigenvalues (ties: higher index first)
      int above = 0;
      for (int j = 0; j < n; ++j) { const double lj = dq[j]; above += (lj > lam) || (lj == lam && j > k); }
      if (above >= eig_rank) lp = 0.0;
    }
    vv[k] = lp;
  }
  Gp::sync();
  for (int b = rank; b < n; b += GS) {
    const double* rb = M + b * ld;
    double* ocol = out + (long long)b * (b + 1) / 2;
    for (int a = 0; a <= b; ++a) {
      const double* ra = M + a * ld;
      double acc0 = 0.0, acc1 = 0.0;
      int k = 0;
      for (; k + 1 < n; k += 2) {
        acc0 += (ra[k] * vv[k]) * rb[k];
        acc1 += (ra[k + 1] * vv[k + 1]) * rb[k + 1];
      }
      if (k < n) acc0 += (ra[k] * vv[k]) * rb[k];
      const double v = acc0 + acc1;
      ocol[a] = (a == b) ? v : v * kSqrt2;
    }
  }
}

// Eigenpairs sorted ascending, eigenvectors column-major (the contract of cusolver.h:76-95).
template <class Gp>
__device__ __forceinline__ void write_sorted_eig(const double* __restrict__ M, const int ld, const int n, const double* dq,
                                 double* __restrict__ Vout, double* __restrict__ Wout) {
  constexpr int GS = Gp::kSize;
  for (int k = Gp::rank(); k < n; k += GS) {
    const double lam = dq[k];
    int pos = 0;
    for (int j = 0; j < n; ++j) {
      const double lj = dq[j];
      pos += (lj < lam) || (lj == lam && j < k);
    }
    Wout[pos] = lam;
    for (int r = 0; r < n; ++r) Vout[(long long)pos * n + r] = M[r * ld + k];
  }
}

}  // namespace cuadmm
