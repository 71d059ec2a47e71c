// Register-resident PSD projection for blocks of size n <= NMAX (NMAX = 4, 8, 16, 32).
//
// NMAX lanes of a wavefront own one block (64/NMAX blocks per wavefront).  Lane r keeps row r of the
// working matrix A and row r of the accumulated orthogonal factor Z in VGPRs (2*NMAX doubles, every
// index a compile-time constant); LDS only carries what lanes must share: the staged input/output
// tile, the current Householder vectors (broadcast reads), and the tridiagonal (d, e).
//
//   load      coalesced svec read -> LDS tile -> rows into registers
//   tridiag   Householder steps k = 0..n-3.  The register row is ROTATED one column per step so that
//             the pivot column is always ar[0]; the loop body is then independent of k and needs no
//             unrolling over k.  Z <- Z*H_k is accumulated in the same step (forward accumulation,
//             row-owner layout, no cross-lane traffic besides the broadcast of v).
//   QL        implicit QL with Wilkinson shift.  One "sweep" walks the unrolled slots i = NMAX-2..0;
//             a slot is active for a block when l <= i < m.  The scalar recurrence runs redundantly in
//             the block's lanes; the plane rotation is applied to the lane's own Z row in registers.
//   rebuild   T = Z*diag(max(d,0)) through LDS, P = T*Z^T row by row written back in place, then a
//             coalesced svec store with the sqrt(2) scaling.
// Arithmetic twin on the CPU: oracle/eigproj_twin.c.
#pragma once
#include <hip/hip_runtime.h>

#include "psd_device.h"

namespace cuadmm {

template <int NMAX>
struct RegLayout {
  static constexpr int LD = NMAX + 1;
  static constexpr int kTile = NMAX * LD;      // staged matrix / reflector-free scratch / T / P
  static constexpr int kVec = 2 * NMAX;        // vv and ww, zero padded for the shifted reads vv[k + j]
  static constexpr int kDE = NMAX + 1;
  static constexpr int kPer = kTile + 2 * kVec + 2 * kDE;
};

template <int NMAX, int MODE, class Args>
__device__ __forceinline__ void psd_small_reg_body(const Args& a, double* smem) {
  using Gp = SubGroup<NMAX>;
  using Lay = RegLayout<NMAX>;
  constexpr int LD = Lay::LD;
  constexpr int BPW = 64 / NMAX;
  const int lane = lane_id();
  const int slot0 = (int)blockIdx.x * BPW;
  long long* dbg = a.dbg ? a.dbg + (long long)blockIdx.x * 8 : nullptr;
#define CUADMM_STAMP(i) do { if (dbg && lane == 0) dbg[i] = (long long)__builtin_readcyclecounter(); } while (0)
  CUADMM_STAMP(0);

  // ---- cooperative coalesced load of the wavefront's blocks into their LDS tiles ---------------
  for (int gg = 0; gg < BPW; ++gg) {
    const int slot = slot0 + gg;
    if (slot >= a.count) break;
    const int bi = a.ids ? a.ids[slot] : slot;
    double* Tg = smem + gg * Lay::kPer;
    if (MODE == 0) {
      const int n = a.bn[bi];
      const double* src = a.in + a.boff[bi];
      const int len = n * (n + 1) / 2;
      for (int e = lane; e < len; e += 64) {
        int i, j;
        tri_decode(e, i, j);
        double v = src[e];
        if (i != j) v *= kSqrt2Inv;
        Tg[j * LD + i] = v;
        Tg[i * LD + j] = v;
      }
    } else {
      const int n = a.n_uniform;
      const double* src = a.in + (long long)bi * n * n;
      for (int idx = lane; idx < n * n; idx += 64) {
        const int c = idx / n, r = idx - c * n;
        if (r >= c) {
          const double v = src[idx];
          Tg[r * LD + c] = v;
          Tg[c * LD + r] = v;
        }
      }
    }
  }
  wave_fence();

  const int g = lane / NMAX;
  const int rank = lane & (NMAX - 1);
  const int slot = slot0 + g;
  if (slot < a.count) {
  const int bi = a.ids ? a.ids[slot] : slot;
  const int n = (MODE == 0) ? a.bn[bi] : a.n_uniform;
  double* T = smem + g * Lay::kPer;
  double* vv = T + Lay::kTile;
  double* ww = vv + Lay::kVec;
  double* D = ww + Lay::kVec;
  double* E = D + Lay::kDE;
  const int half_base = lane & ~(NMAX - 1);

  double ar[NMAX], q[NMAX];
#pragma unroll
  for (int c = 0; c < NMAX; ++c) {
    ar[c] = (rank < n && c < n) ? T[rank * LD + c] : 0.0;
    q[c] = (c == rank) ? 1.0 : 0.0;
  }
  vv[rank] = 0.0; vv[rank + NMAX] = 0.0;
  ww[rank] = 0.0; ww[rank + NMAX] = 0.0;
  wave_fence();
  CUADMM_STAMP(1);

  // ---- Householder tridiagonalisation, Z accumulated on the fly ---------------------------------
  for (int k = 0; k < n - 2; ++k) {
    const double x = ar[0];                                    // A[rank][k]
    const double xn2 = Gp::sum((rank >= k + 2) ? x * x : 0.0, nullptr);
    const double alpha = __shfl(x, half_base + k + 1, 64);
    if (rank == k) D[k] = x;
    double t = 0.0, beta = alpha, scal = 0.0;
    if (xn2 != 0.0) {
      double nrm, inrm;
      fast_sqrt_rsqrt(alpha * alpha + xn2, nrm, inrm);
      beta = -copysign(nrm, alpha);
      t = (beta - alpha) * (-copysign(inrm, alpha));          // (beta - alpha) / beta
      scal = fast_rcp(alpha - beta);
    }
    if (rank == k + 1) E[k] = beta;
    if (t != 0.0) {
      const double vr = (rank >= k + 2) ? x * scal : ((rank == k + 1) ? 1.0 : 0.0);
      vv[rank] = vr;
      wave_fence();
      // p = t * A(k+1:, k+1:) v   (ar[j] = A[rank][k+j]; vv[c] = 0 for c <= k)
      double p0 = 0.0, p1 = 0.0;
      const double* vs = vv + k;
#pragma unroll
      for (int j = 0; j < NMAX; j += 2) { p0 += ar[j] * vs[j]; p1 += ar[j + 1] * vs[j + 1]; }
      const double pr = (rank >= k + 1) ? (p0 + p1) * t : 0.0;
      const double K = Gp::sum(pr * vr, nullptr) * (-0.5 * t);
      const double wr = pr + K * vr;
      ww[rank] = wr;
      // Z <- Z H_k  (row-owner: s = t * <z_row, v>)
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int c = 0; c < NMAX; c += 2) { s0 += q[c] * vv[c]; s1 += q[c + 1] * vv[c + 1]; }
      const double sr = (s0 + s1) * t;
#pragma unroll
      for (int c = 0; c < NMAX; ++c) q[c] -= sr * vv[c];
      wave_fence();
      const double* wsft = ww + k;
#pragma unroll
      for (int j = 0; j < NMAX; ++j) ar[j] -= vr * wsft[j] + wr * vs[j];
      wave_fence();
    }
#pragma unroll
    for (int j = 0; j + 1 < NMAX; ++j) ar[j] = ar[j + 1];
    ar[NMAX - 1] = 0.0;
  }
  {
    const int kk = n >= 2 ? n - 2 : 0;                          // columns already rotated out
    if (n >= 2) {
      if (rank == kk) D[kk] = ar[0];
      if (rank == kk + 1) { E[kk] = ar[0]; D[kk + 1] = ar[1]; }
    } else if (rank == 0) {
      D[0] = ar[0];
    }
    if (rank == 0) E[n - 1] = 0.0;
  }
  wave_fence();

  CUADMM_STAMP(2);
  // deflation threshold relative to ||T||_F (see psd_device.h)
  double eps_abs;
  {
    const double dv = rank < n ? D[rank] : 0.0, ev = rank < n ? E[rank] : 0.0;
    eps_abs = sqrt(Gp::sum(dv * dv + 2.0 * ev * ev, nullptr)) * 0x1p-53;
  }

  // ---- implicit QL ---------------------------------------------------------------------------------
  int l = 0, m = 0, sweeps = 0, fail = 0;
  bool done = (n <= 1);
  const bool writer = (rank == 0);
  while (!done) {
    for (;;) {
      m = Gp::first_true(l, n, [&](int idx) {
        if (idx >= n - 1) return true;
        const double ae = fabs(E[idx]);
        const double dd = fabs(D[idx]) + fabs(D[idx + 1]);
        return ae <= eps_abs || ae + dd == dd;
      });
      if (m >= n) m = n - 1;
      if (m > l) break;
      ++l; sweeps = 0;
      if (l >= n) { done = true; break; }
    }
    if (done) break;
    if (sweeps++ >= kQlMaxSweepsPerEig) { fail = 1; break; }
    const double dl = D[l], el = E[l];
    double gq = (D[l + 1] - dl) * fast_rcp(el + el);
    double r0, r0i;
    fast_sqrt_rsqrt(fma(gq, gq, 1.0), r0, r0i);
    gq = D[m] - dl + el * fast_rcp(gq + copysign(r0, gq));
    double s = 1.0, c = 1.0, p = 0.0;
    double e_c = E[m - 1], d_c = D[m - 1], d1_c = D[m];      // operands of the first active slot i = m-1
    // Slot predicates depend only on (i, l, m): nothing on the serial critical path feeds a branch (a data
    // dependent exit test costs ~75 cycles per slot on gfx950, tools/ubench/loop_overheads.hip).  The
    // textbook's "r == 0" underflow exit cannot trigger for blocks whose norm is in the normal fp64 range
    // (every off-diagonal inside the window exceeds eps*||T||); should it happen anyway the block is
    // flagged as failed instead of being special-cased.
#pragma unroll
    for (int i = NMAX - 2; i >= 0; --i) {
      if (i < m && i >= l) {
        double e_n = 0.0, d_n = 0.0;
        if (i > 0) { e_n = E[i - 1]; d_n = D[i - 1]; }          // prefetch for slot i-1 (not yet touched this sweep)
        const double f = s * e_c, b = c * e_c;
        const double h = fma(f, f, gq * gq);
        fail |= (h == 0.0) ? 2 : 0;
        double rr, rinv;
        fast_sqrt_rsqrt(h, rr, rinv);
        s = f * rinv; c = gq * rinv;
        gq = d1_c - p;
        const double cb = c * b;
        const double r2 = fma(d_c - gq, s, cb + cb);
        p = s * r2;
        const double dnew = gq + p;
        gq = fma(c, r2, -b);
        const double z0 = q[i], z1 = q[i + 1];
        q[i + 1] = fma(s, z0, c * z1);
        q[i] = fma(c, z0, -(s * z1));
        if (writer) { E[i + 1] = rr; D[i + 1] = dnew; }
        d1_c = d_c; e_c = e_n; d_c = d_n;
      }
    }
    const bool broke = false;
    if (!broke && writer) { D[l] = D[l] - p; E[l] = gq; E[m] = 0.0; }
    wave_fence();
  }

  CUADMM_STAMP(3);
  // ---- output ------------------------------------------------------------------------------------------
  if (MODE == 0) {
    // lambda+ = max(d,0) (dense_scalar.cu:41-47); with a rank limit only the a.eig_rank LARGEST eigenvalues keep their
    // positive part: max(W,0) * mask with mask = 1 on the last eig_rank entries of the ascending spectrum
    // (dense_scalar.cu:51-57, get_eig_rank_mask.cu:13-37).  D is unsorted here, so "among the eig_rank largest" is
    // decided by counting the eigenvalues above (ties: higher index first).
    {
      const double lam = rank < n ? D[rank] : 0.0;
      double lp = lam > 0.0 ? lam : 0.0;
      if (a.eig_rank > 0 && rank < n) {
        int above = 0;
        for (int j = 0; j < n; ++j) { const double lj = D[j]; above += (lj > lam) || (lj == lam && j > rank); }
        if (above >= a.eig_rank) lp = 0.0;
      }
      vv[rank] = lp;
    }
    wave_fence();
    if constexpr (NMAX >= 16) {
      // hand Z (unscaled) and lambda+ (in vv) to the matrix-core rebuild that follows (psd_small_reg_rebuild_mfma)
#pragma unroll
      for (int k = 0; k < NMAX; ++k) T[rank * LD + k] = q[k];
    } else {
      // T = Z * diag(lambda+)   (diagonal_batch.cu:11-23)
#pragma unroll
      for (int k = 0; k < NMAX; ++k) T[rank * LD + k] = q[k] * vv[k];
      wave_fence();
      // P = T * Z^T, upper triangle, row a at a time; row a of T is dead once it has been used
      for (int aa = 0; aa < n; ++aa) {
        const double* ta = T + aa * LD;
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int k = 0; k < NMAX; k += 2) { acc0 += ta[k] * q[k]; acc1 += ta[k + 1] * q[k + 1]; }
        T[aa * LD + rank] = acc0 + acc1;                      // P[aa][rank]
      }
    }
    wave_fence();
    if (fail && rank == 0 && a.info) atomicAdd(a.info, 1);
  } else {
    // ascending order + column-major eigenvectors (cusolver.h:76-95)
    int pos = 0;
    const double lam = rank < n ? D[rank] : 0.0;
    for (int j = 0; j < n; ++j) {
      const double lj = D[j];
      pos += (lj < lam) || (lj == lam && j < rank);
    }
    int* POS = reinterpret_cast<int*>(vv);
    if (rank < n) {
      POS[rank] = pos;
      a.Wout[(long long)bi * n + pos] = lam;
    }
    wave_fence();
    if (rank < n) {
      double* Vout = a.out + (long long)bi * n * n;
#pragma unroll
      for (int k = 0; k < NMAX; ++k)
        if (k < n) Vout[(long long)POS[k] * n + rank] = q[k];
    }
    if (rank == 0 && a.info) a.info[bi] = fail;
  }
  CUADMM_STAMP(4);
  }  // slot < count
#undef CUADMM_STAMP
}

// P = (Z diag(l+)) Z^T on the fp64 matrix cores for the blocks of one wavefront (NMAX = 16 or 32), executed
// by the whole wavefront in uniform control flow after every group has left the eigen-solver.
// v_mfma_f64_16x16x4_f64: lane l feeds A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; the four results of
// a lane are D[row = (l>>4) + 4*reg][col = l&15].  A = Z*diag(l+) is formed on the fly from the LDS tile
// (the reference's diagonal_batch.cu:11-23 + cublas.h:18-35 in one step).  Only the tiles of the upper
// triangle are computed (svec stores the upper triangle).
typedef double psd_v4f64 __attribute__((ext_vector_type(4)));
template <int NMAX, class Lay, class Args>
__device__ __forceinline__ void psd_small_reg_rebuild_mfma(const Args& a, double* smem, const int slot0) {
  constexpr int LD = Lay::LD;
  constexpr int BPW = 64 / NMAX;
  constexpr int NT = NMAX >= 16 ? NMAX / 16 : 1;  // tiles per dimension
  const int lane = lane_id();
  const int r16 = lane & 15, kk = lane >> 4;
  for (int gg = 0; gg < BPW; ++gg) {
    if (slot0 + gg >= a.count) break;
    double* T = smem + gg * Lay::kPer;
    const double* lam = T + Lay::kTile;         // vv: max(d,0)
    psd_v4f64 acc[NT * (NT + 1) / 2];
#pragma unroll
    for (int t = 0; t < NT * (NT + 1) / 2; ++t) acc[t] = psd_v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k0 = 0; k0 < NMAX; k0 += 4) {
      const int k = k0 + kk;
      const double lk = lam[k];
      double zf[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) zf[t] = T[(16 * t + r16) * LD + k];
      int idx = 0;
#pragma unroll
      for (int ti = 0; ti < NT; ++ti)
#pragma unroll
        for (int tj = ti; tj < NT; ++tj, ++idx)
          acc[idx] = __builtin_amdgcn_mfma_f64_16x16x4f64(zf[ti] * lk, zf[tj], acc[idx], 0, 0, 0);
    }
    wave_fence();                               // every lane has finished reading Z
    int idx = 0;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti)
#pragma unroll
      for (int tj = ti; tj < NT; ++tj, ++idx)
#pragma unroll
        for (int r = 0; r < 4; ++r) T[(16 * ti + kk + 4 * r) * LD + 16 * tj + r16] = acc[idx][r];
  }
  wave_fence();
}

// store phase, executed by the whole wavefront after every group has finished
template <int NMAX, class Lay, class Args>
__device__ __forceinline__ void psd_small_reg_store(const Args& a, const double* smem, const int slot0, long long* dbg) {
  constexpr int LD = Lay::LD;
  constexpr int BPW = 64 / NMAX;
  const int lane = lane_id();
  for (int gg = 0; gg < BPW; ++gg) {
    const int slot = slot0 + gg;
    if (slot >= a.count) break;
    const int bi = a.ids ? a.ids[slot] : slot;
    const double* Tg = smem + gg * Lay::kPer;
    const int n = a.bn[bi];
    double* dst = a.out + a.boff[bi];
    const int len = n * (n + 1) / 2;
    for (int e = lane; e < len; e += 64) {
      int i, j;
      tri_decode(e, i, j);                       // column i, row j <= i
      const double v = Tg[j * LD + i];
      dst[e] = (i == j) ? v : v * kSqrt2;
    }
  }
  if (dbg && lane == 0) dbg[5] = (long long)__builtin_readcyclecounter();
}

}  // namespace cuadmm
