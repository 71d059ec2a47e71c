// PSD projection of blocks with 32 < n <= 64: the matrix-sign iteration of psd_large.hip with the WHOLE iteration
// resident in LDS -- one workgroup (6 or 10 wavefronts) per block, two NP x NP matrices (S, S^2) in LDS and the next S in
// accumulator registers, one launch, no global traffic between the svec read and the svec write.
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
#include "sign_sched.h"
#include "wave_reduce.h"

namespace cuadmm {

typedef double sl_v4f64 __attribute__((ext_vector_type(4)));
typedef double sl_v2f64 __attribute__((ext_vector_type(2)));
typedef int sl_v4i32 __attribute__((ext_vector_type(4)));


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

// acc = A * B on the wavefront's upper sub-tile (ti, tj).  A symmetric (read as A[k][row]).  Operands in LDS.
// kmax: the k-steps beyond the block's true size (a multiple of 8 >= n) multiply rows of zeros -- the padding of the tile -- and are skipped: the
// kernel is bound by the matrix pipe of the SIMDs that carry three of the ten wavefronts (3 x 16 x 64 cycles per product at NP = 64: the measured
// 6 k cycles per step), so n = 55 (pendulum's 80 blocks) gets an eighth of its products back (159 -> 149 us).  Same bits: what is skipped adds 0 * 0.
// Measured and rejected (round 6): twelve wavefronts, the last two sub-tiles split in halves of the k range so that every SIMD carries two and a
// half (35 instead of 42 MFMAs per product), the second half's accumulators handed over through LDS behind a flag: 149 -> 175 us.
template <int NP>
__device__ __forceinline__ sl_v4f64 sl_mma(const double* __restrict__ A, const double* __restrict__ B, int ti, int tj, int lane, int kmax = NP) {
  constexpr int LD = SignLdsCfg<NP>::LD;
  const int r16 = lane & 15, kk = lane >> 4;
  const double* arow = A + kk * LD + ti * 16 + r16;
  const double* brow = B + kk * LD + tj * 16 + r16;
  sl_v4f64 acc = sl_v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k0 = 0; k0 < NP; k0 += 8) {
    if (k0 >= kmax) break;                   // wave-uniform
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[k0 * LD], brow[k0 * LD], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[(k0 + 4) * LD], brow[(k0 + 4) * LD], acc, 0, 0, 0);
  }
  return acc;
}
// the sub-tile (ti, tj) of a matrix in LDS, in accumulator layout
template <int NP>
__device__ __forceinline__ sl_v4f64 sl_tile(const double* __restrict__ E, int ti, int tj, int lane) {
  constexpr int LD = SignLdsCfg<NP>::LD;
  const int r16 = lane & 15, kk = lane >> 4;
  sl_v4f64 v;
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = E[(ti * 16 + kk + 4 * r) * LD + tj * 16 + r16];
  return v;
}
// mirrored store of the sub-tile (ti, tj); on a diagonal sub-tile the upper triangle decides
template <int NP>
__device__ __forceinline__ void sl_store(double* __restrict__ C, const sl_v4f64& v, int ti, int tj, int lane) {
  constexpr int LD = SignLdsCfg<NP>::LD;
  const int r16 = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = ti * 16 + kk + 4 * r, col = tj * 16 + r16;
    if (ti == tj && col < row) continue;
    C[row * LD + col] = v[r];
    C[col * LD + row] = v[r];
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

// Per-block adaptive schedule (sign_sched.h): every wavefront reduces the statistics of its own sub-tile (off-diagonal
// sub-tiles count twice), the workgroup sums the NU partials in a fixed order after the barrier that the products need
// anyway (+1 barrier per step between S Y and the combine), and every thread runs the same (uniform) state machine.
// TRIPLE: a third matrix in LDS (the next iterate is stored beside the current one: no barrier between the last read of S and the
// store of T) and the statistics -- three wave reductions, their exchange and its barrier -- only on the steps whose decision reads
// them (SignSched::needs_stats: same decisions, same iterates bit for bit).  Two barriers per lift / probe step instead of three;
// a moment relaxation's blocks spend 35 of ~41 steps there.  Taken when the class has at most one workgroup per CU anyway.
template <int NP, bool TRIPLE>
__device__ __forceinline__ void psd_sign_lds_body(const double* __restrict__ in, double* __restrict__ out, int n, int* fail,
                                                  double* smem, int* steps_out, int* hint) {
  using Cfg = SignLdsCfg<NP>;
  constexpr int LD = Cfg::LD;
  // TWO matrices in LDS (S and Y); the next iterate T lives in the accumulator registers of the wavefront that owns the
  // sub-tile and overwrites S in place once every wavefront has finished reading S (the barrier the statistics exchange
  // needs anyway).  Three resident matrices (round 1) limited a CU to 2 (NP = 48) / 1 (NP = 64) workgroups; two allow
  // 4 / 2, and the exposed LDS / barrier latency of one block is covered by the others (the kernel ran at ~30 % of the
  // fp64 matrix-core peak on the n = 45 blocks of C4).
  double* S = smem;
  double* Y = S + NP * LD;
  double* S2 = TRIPLE ? Y + NP * LD : S;
  __shared__ double red[64];
  __shared__ double stat[3 * 16];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kk = lane >> 4;
  int ti, tj;
  sl_upper_tile<Cfg::NT>(wave, ti, tj);
  const double wgt = ti == tj ? 1.0 : 2.0;
  const int kmax = __builtin_amdgcn_readfirstlane((n + 7) & ~7);
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
  if (TRIPLE) for (int e = tid; e < NP * LD; e += Cfg::THREADS) S2[e] = 0.0;       // the padding of the second copy
  __syncthreads();
  SignSched sched;
  if (hint && *hint > 0) sched.lift0 = *hint;
  bool last = false;
  while (!last) {
    const bool stats = !TRIPLE || sched.needs_stats();
    const sl_v4f64 y = sl_mma<NP>(S, S, ti, tj, lane, kmax);                               // Y = S*S
    sl_store<NP>(Y, y, ti, tj, lane);
    double pa = 0.0, pb = 0.0;
    if (stats) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (ti == tj && kk + 4 * r == r16) pa += y[r];
        pb += y[r] * y[r];
      }
      pa = wave_sum(pa);
      pb = wave_sum(pb) * wgt;
    }
    __syncthreads();
    const sl_v4f64 z = sl_mma<NP>(S, Y, ti, tj, lane, kmax);                               // S*Y
    const sl_v4f64 e = sl_tile<NP>(S, ti, tj, lane);
    double mu;
    if (stats) {
      double pg = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) { const double d = e[r] - z[r]; pg += d * d; }
      pg = wave_sum(pg) * wgt;
      if (lane == 0) { stat[wave] = pa; stat[16 + wave] = pb; stat[32 + wave] = pg; }
      __syncthreads();                                                               // statistics visible; all reads of S done
      double ta = 0.0, tb = 0.0, tg = 0.0;
#pragma unroll
      for (int w = 0; w < Cfg::NU; ++w) { ta += stat[w]; tb += stat[16 + w]; tg += stat[32 + w]; }
      mu = sched.decide<false>(n, ta, tb, tg, last);
    } else {
      mu = sched.decide<false>(n, 0.0, 0.0, 0.0, last);                              // the scale of this step was fixed in advance
    }
    double alpha, beta;
    sched.coefs(mu, alpha, beta);
    sl_v4f64 t;
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = fma(alpha, z[r], beta * e[r]);                     // T = 1.5 mu S - 0.5 mu^3 S*Y
    sl_store<NP>(S2, t, ti, tj, lane);                                               // the sub-tile and its mirror image (TRIPLE: beside S, else in place)
    __syncthreads();
    if (TRIPLE) { double* u = S; S = S2; S2 = u; }
  }
  if (steps_out && tid == 0) *steps_out = sched.steps;
  if (hint && tid == 0) *hint = sched.lifts;
  // P = 0.5 * (X0 + X0 * S): X0 is unpacked again into Y; the result replaces S after a barrier
  sl_unpack<NP>(in, n, Y, tid);
  sl_v4f64 t;
  {
    const sl_v4f64 z = sl_mma<NP>(Y, S, ti, tj, lane, kmax);
    const sl_v4f64 e = sl_tile<NP>(Y, ti, tj, lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = 0.5 * z[r] + 0.5 * e[r];
  }
  __syncthreads();
  sl_store<NP>(S, t, ti, tj, lane);
  __syncthreads();
  const int len = n * (n + 1) / 2;
  bool bad = false;
  for (int e = tid; e < len; e += Cfg::THREADS) {
    int i, j;
    tri_decode(e, i, j);
    const double v = S[j * LD + i];
    bad |= !(fabs(v) <= 1.7976931348623157e308);
    out[e] = (i == j) ? v : v * kSqrt2;
  }
  if (bad && fail) atomicAdd(fail, 1);
}

}  // namespace cuadmm
