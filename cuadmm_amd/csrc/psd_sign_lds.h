// PSD projection of blocks with 32 < n <= 64: the matrix-sign iteration of psd_large.hip with the WHOLE iteration
// resident in LDS -- one workgroup (6 or 10 wavefronts) per block, three NP x NP matrices (S, S^2, next S) in LDS, one
// launch, no global traffic between the svec read and the svec write.
//
// Why not the register-resident eigensolver (psd_small_reg.h) that serves n <= 32: at n = 33..64 it runs one block
// per wavefront and is bound by the dependent rotation chain (0.7 us/block at n = 45 in bulk, 0.84 ms latency for a
// single n = 55 block).  The sign iteration is 89 small GEMMs on v_mfma_f64_16x16x4_f64 with operands read straight
// from LDS: measured 3x the throughput and 4x lower latency.
//
// Layout: odd row stride LD = NP + 1: both the fragment reads (16 consecutive doubles of 4 k-rows) and the mirrored
// (transposed) stores of the epilogue are then (almost) bank-conflict free; LD = 16 mod 32 is perfect for the reads
// but makes the transposed stores 8-way conflicting.  Only the 16 x 16 sub-tiles on or above the diagonal are
// computed (6 of 9 / 10 of 16), one wavefront each, and mirrored on store: fewer MFMAs, exactly symmetric iterate.
#pragma once
#include <hip/hip_runtime.h>

#include "psd_device.h"

namespace cuadmm {

typedef double sl_v4f64 __attribute__((ext_vector_type(4)));

template <int NP> struct SignLdsCfg;
// one wavefront per upper sub-tile (NU wavefronts per workgroup): with a single wavefront per SIMD the LDS latency of
// every k-step is exposed (measured: 25 k cycles per Newton-Schulz step instead of 6 k)
template <> struct SignLdsCfg<48> { static constexpr int LD = 49, NT = 3, NU = 6, THREADS = 64 * 6; };
template <> struct SignLdsCfg<64> { static constexpr int LD = 65, NT = 4, NU = 10, THREADS = 64 * 10; };

// q-th upper sub-tile (row-major over i <= j) -> (i, j)
template <int NT>
__device__ __forceinline__ void sl_upper_tile(int q, int& i, int& j) {
  i = 0;
  int rem = q, len = NT;
  while (rem >= len) { rem -= len; --len; ++i; }
  j = i + rem;
}

// C = alpha * A * B + beta * E on the upper sub-tiles, mirrored.  A symmetric (read as A[k][row]).  All in LDS.
// Wavefront w owns the w-th upper sub-tile (ti, tj).
template <int NP>
__device__ __forceinline__ void sl_gemm(const double* __restrict__ A, const double* __restrict__ B, const double* __restrict__ E,
                                        double alpha, double beta, double* __restrict__ C, int ti, int tj, int lane) {
  using Cfg = SignLdsCfg<NP>;
  constexpr int LD = Cfg::LD;
  const int r16 = lane & 15, kk = lane >> 4;
  const double* arow = A + kk * LD + ti * 16 + r16;
  const double* brow = B + kk * LD + tj * 16 + r16;
  sl_v4f64 acc = sl_v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k0 = 0; k0 < NP; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[k0 * LD], brow[k0 * LD], acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = ti * 16 + kk + 4 * r, col = tj * 16 + r16;
    if (ti == tj && col < row) continue;      // diagonal sub-tile: the upper triangle decides
    double v = alpha * acc[r];
    if (E) v += beta * E[row * LD + col];
    C[row * LD + col] = v;
    C[col * LD + row] = v;
  }
}

template <int NP>
__device__ __forceinline__ void sl_unpack(const double* __restrict__ src, int n, double* __restrict__ M, int tid) {
  constexpr int LD = SignLdsCfg<NP>::LD;
  for (int e = tid; e < NP * LD; e += SignLdsCfg<NP>::THREADS) M[e] = 0.0;
  __syncthreads();
  const int len = n * (n + 1) / 2;
  for (int e = tid; e < len; e += SignLdsCfg<NP>::THREADS) {
    int i, j;
    tri_decode(e, i, j);
    double v = src[e];
    if (i != j) v *= kSqrt2Inv;
    M[j * LD + i] = v;
    M[i * LD + j] = v;
  }
  __syncthreads();
}

// kLift / kPolish / kMu: the schedule of psd_large.hip (SignPsd)
template <int NP>
__device__ __forceinline__ void psd_sign_lds_body(const double* __restrict__ in, double* __restrict__ out, int n, int* fail,
                                                  double* smem, int lift_steps, int polish_steps, double lift_mu) {
  using Cfg = SignLdsCfg<NP>;
  constexpr int LD = Cfg::LD;
  double* S = smem;
  double* Y = S + NP * LD;
  double* T = Y + NP * LD;
  __shared__ double red[64];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int ti, tj;
  sl_upper_tile<Cfg::NT>(wave, ti, tj);
  sl_unpack<NP>(in, n, S, tid);
  // ||X||_1 = max column sum (symmetric: row sums), S <- X / ||X||_1
  if (tid < 64) {
    double s = 0.0;
    if (tid < NP)
      for (int r = 0; r < NP; ++r) s += fabs(S[r * LD + tid]);
    red[tid] = s;
  }
  __syncthreads();
  double nrm = 0.0;
  for (int c = 0; c < NP; ++c) { const double v = red[c]; nrm = (v > nrm || !(v == v)) ? v : nrm; }
  const double scale = nrm > 0.0 ? 1.0 / nrm : (nrm == 0.0 ? 0.0 : nrm);
  for (int e = tid; e < NP * LD; e += Cfg::THREADS) S[e] *= scale;
  __syncthreads();
  for (int it = 0; it < lift_steps + polish_steps; ++it) {
    const double mu = it < lift_steps ? lift_mu : 1.0;
    sl_gemm<NP>(S, S, nullptr, 1.0, 0.0, Y, ti, tj, lane);                          // Y = S*S
    __syncthreads();
    sl_gemm<NP>(S, Y, S, -0.5 * mu * mu * mu, 1.5 * mu, T, ti, tj, lane);           // T = 1.5 mu S - 0.5 mu^3 S*Y
    __syncthreads();
    double* t = S; S = T; T = t;
  }
  // P = 0.5 * (X0 + X0 * S): X0 is unpacked again (three matrices fit in LDS, four do not at NP = 64)
  sl_unpack<NP>(in, n, Y, tid);
  sl_gemm<NP>(Y, S, Y, 0.5, 0.5, T, ti, tj, lane);
  __syncthreads();
  const int len = n * (n + 1) / 2;
  bool bad = false;
  for (int e = tid; e < len; e += Cfg::THREADS) {
    int i, j;
    tri_decode(e, i, j);
    const double v = T[j * LD + i];
    bad |= !(fabs(v) <= 1.7976931348623157e308);
    out[e] = (i == j) ? v : v * kSqrt2;
  }
  if (bad && fail) atomicAdd(fail, 1);
}

}  // namespace cuadmm
