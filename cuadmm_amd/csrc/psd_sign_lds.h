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
template <int NP>
__device__ __forceinline__ sl_v4f64 sl_mma(const double* __restrict__ A, const double* __restrict__ B, int ti, int tj, int lane) {
  constexpr int LD = SignLdsCfg<NP>::LD;
  const int r16 = lane & 15, kk = lane >> 4;
  const double* arow = A + kk * LD + ti * 16 + r16;
  const double* brow = B + kk * LD + tj * 16 + r16;
  sl_v4f64 acc = sl_v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k0 = 0; k0 < NP; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(arow[k0 * LD], brow[k0 * LD], acc, 0, 0, 0);
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
template <int NP>
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
  __shared__ double red[64];
  __shared__ double stat[3 * 16];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, kk = lane >> 4;
  int ti, tj;
  sl_upper_tile<Cfg::NT>(wave, ti, tj);
  const double wgt = ti == tj ? 1.0 : 2.0;
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
  SignSched sched;
  if (hint && *hint > 0) sched.lift0 = *hint;
  bool last = false;
  while (!last) {
    const sl_v4f64 y = sl_mma<NP>(S, S, ti, tj, lane);                               // Y = S*S
    sl_store<NP>(Y, y, ti, tj, lane);
    double pa = 0.0, pb = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (ti == tj && kk + 4 * r == r16) pa += y[r];
      pb += y[r] * y[r];
    }
    pa = wave_sum(pa);
    pb = wave_sum(pb) * wgt;
    __syncthreads();
    const sl_v4f64 z = sl_mma<NP>(S, Y, ti, tj, lane);                               // S*Y
    const sl_v4f64 e = sl_tile<NP>(S, ti, tj, lane);
    double pg = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const double d = e[r] - z[r]; pg += d * d; }
    pg = wave_sum(pg) * wgt;
    if (lane == 0) { stat[wave] = pa; stat[16 + wave] = pb; stat[32 + wave] = pg; }
    __syncthreads();                                                                 // statistics visible; all reads of S done
    double ta = 0.0, tb = 0.0, tg = 0.0;
#pragma unroll
    for (int w = 0; w < Cfg::NU; ++w) { ta += stat[w]; tb += stat[16 + w]; tg += stat[32 + w]; }
    const double mu = sched.decide<false>(n, ta, tb, tg, last);
    const double alpha = -0.5 * mu * mu * mu, beta = 1.5 * mu;
    sl_v4f64 t;
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = alpha * z[r] + beta * e[r];                   // T = 1.5 mu S - 0.5 mu^3 S*Y
    sl_store<NP>(S, t, ti, tj, lane);                                                // in place: the sub-tile and its mirror image
    __syncthreads();
  }
  if (steps_out && tid == 0) *steps_out = sched.steps;
  if (hint && tid == 0) *hint = sched.lifts;
  // P = 0.5 * (X0 + X0 * S): X0 is unpacked again into Y; the result replaces S after a barrier
  sl_unpack<NP>(in, n, Y, tid);
  sl_v4f64 t;
  {
    const sl_v4f64 z = sl_mma<NP>(Y, S, ti, tj, lane);
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

// ---------------------------------------------------------------------------------------------------------------
// n <= 32: one WAVEFRONT per block, no workgroup barrier.  S lives in LDS (32 x 33 doubles per wavefront) only to be
// re-read in MFMA operand layout; everything else stays in registers:
//   * the accumulator layout of v_mfma_f64_16x16x4_f64 (lane (kk, c), register r holds element (kk + 4 r, c)) IS the
//     B-operand layout of k-step r, so Y = S^2 never leaves the registers: its four 16 x 16 sub-tiles feed the second
//     product directly;
//   * the 16 operand fragments of S read for Y = S S (8 k-steps x 2 column halves) are the A fragments of S (S Y) too
//     (S symmetric), so the second product issues no LDS reads;
//   * T = 1.5 mu S - 0.5 mu^3 S Y is formed on the three upper sub-tiles from the register copy of S in accumulator
//     layout and written back to LDS mirrored (exact symmetry).
// Per step: 16 + 12 + 4 LDS reads, <= 24 + 4 LDS writes, 48 MFMAs (24 for the upper sub-tiles of Y, 24 for S Y; the
// lower sub-tile of Y is transposed through LDS).
// ---------------------------------------------------------------------------------------------------------------
struct SignWave32 {
  static constexpr int NP = 32, LD = 33, SCR_LD = 17;
  static constexpr int PER_WAVE = NP * LD + 16 * SCR_LD;   // S + a 16 x 16 transposition tile
};

__device__ __forceinline__ void sw32_unpack(const double* __restrict__ src, int n, double* __restrict__ M, int lane) {
  constexpr int LD = SignWave32::LD;
  for (int e = lane; e < 32 * LD; e += 64) M[e] = 0.0;
  wave_fence();
  const int len = n * (n + 1) / 2;
  for (int e = lane; e < len; e += 64) {
    int i, j;
    tri_decode(e, i, j);
    double v = src[e];
    if (i != j) v *= kSqrt2Inv;
    M[j * LD + i] = v;
    M[i * LD + j] = v;
  }
  wave_fence();
}

// svec <-> LDS tile by COLUMNS: column c of the upper triangle is the contiguous svec range [c (c + 1) / 2, + c + 1), and
// column c + 1 follows it immediately, so the two half-waves take columns 2 q and 2 q + 1 (lane & 31 = row): 16 coalesced,
// independent loads per lane, no index decoding (tri_decode costs a float sqrt and two correction loops per element; the
// prologue and epilogue were 37 % of a block's lifetime once the adaptive schedule cut the iteration to 12 steps).
template <int NQ>
__device__ __forceinline__ void sw32_load_cols(const double* __restrict__ src, int n, int lane, int q0, double (&v)[NQ]) {
  const int h = lane >> 5, r = lane & 31;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int c = 2 * (q0 + q) + h;
    v[q] = (r <= c && c < n) ? src[c * (c + 1) / 2 + r] : 0.0;
  }
}
// M (32 x 32, stride LD) = scale * smat(v), both triangles; the padding rows / columns (>= n) receive zeros
template <int NQ>
__device__ __forceinline__ void sw32_tile_from_cols(double* __restrict__ M, int lane, int q0, const double (&v)[NQ], double scale) {
  constexpr int LD = SignWave32::LD;
  const int h = lane >> 5, r = lane & 31;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int c = 2 * (q0 + q) + h;
    if (r <= c) {
      const double x = v[q] * (r == c ? scale : scale * kSqrt2Inv);
      M[r * LD + c] = x;
      M[c * LD + r] = x;
    }
  }
}
// svec(out) = upper triangle of M by columns (sqrt2 off the diagonal); returns whether a non-finite value was seen
__device__ __forceinline__ bool sw32_store_cols(const double* __restrict__ M, double* __restrict__ out, int n, int lane) {
  constexpr int LD = SignWave32::LD;
  const int h = lane >> 5, r = lane & 31;
  bool bad = false;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int c = 2 * q + h;
    if (r <= c && c < n) {
      const double x = M[r * LD + c];
      bad |= !(fabs(x) <= 1.7976931348623157e308);
      out[c * (c + 1) / 2 + r] = (r == c) ? x : x * kSqrt2;
    }
  }
  return bad;
}

// all 16 operand fragments of the symmetric matrix in LDS: f[s][x] = M[4 s + kk][16 x + r16]
__device__ __forceinline__ void sw32_frags(const double* __restrict__ M, int r16, int kk, double (&f)[8][2]) {
  constexpr int LD = SignWave32::LD;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    f[s][0] = M[(4 * s + kk) * LD + r16];
    f[s][1] = M[(4 * s + kk) * LD + 16 + r16];
  }
}

// accumulator-layout copy of the three upper sub-tiles: d[0] = (0,0), d[1] = (0,1), d[2] = (1,1)
__device__ __forceinline__ void sw32_dlayout(const double* __restrict__ M, int r16, int kk, sl_v4f64 (&d)[3]) {
  constexpr int LD = SignWave32::LD;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    d[0][r] = M[(kk + 4 * r) * LD + r16];
    d[1][r] = M[(kk + 4 * r) * LD + 16 + r16];
    d[2][r] = M[(16 + kk + 4 * r) * LD + 16 + r16];
  }
}

// mirrored store of the three upper sub-tiles
__device__ __forceinline__ void sw32_store(double* __restrict__ M, int r16, int kk, const sl_v4f64 (&d)[3]) {
  constexpr int LD = SignWave32::LD;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = kk + 4 * r, col = r16;
    if (col >= row) { M[row * LD + col] = d[0][r]; M[col * LD + row] = d[0][r]; }
    M[row * LD + 16 + col] = d[1][r];
    M[(16 + col) * LD + row] = d[1][r];
    if (col >= row) { M[(16 + row) * LD + 16 + col] = d[2][r]; M[(16 + col) * LD + 16 + row] = d[2][r]; }
  }
}

// acc(upper sub-tiles) = A * B with A given by its fragments fa (symmetric A) and B by its four sub-tiles in accumulator
// layout yb[b][c] (row block b, column block c)
__device__ __forceinline__ void sw32_mma_regB(const double (&fa)[8][2], const sl_v4f64 (&yb)[2][2], sl_v4f64 (&acc)[3]) {
#pragma unroll
  for (int t = 0; t < 3; ++t) acc[t] = sl_v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ks = 4 * b + s;   // k rows 16 b + 4 s + kk
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ks][0], yb[b][0][s], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ks][0], yb[b][1][s], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ks][1], yb[b][1][s], acc[2], 0, 0, 0);
    }
}

// Per-block adaptive schedule (sign_sched.h): the statistics come from registers the step already holds -- tr Y and
// ||Y||_F^2 from the accumulators of Y = S S, ||S - S Y||_F^2 from the accumulators of S Y and the copy of S that the
// combine step reads anyway -- three wave reductions per step next to 48 MFMAs.
template <bool DBG>
__device__ __forceinline__ void psd_sign_wave32_body(const double* __restrict__ in, double* __restrict__ out, int n, int* fail,
                                                     double* S, int* steps_out, int* hint, long long* dbg) {
#define SW32_STAMP(k) do { if (DBG) { const long long now_ = (long long)__builtin_readcyclecounter(); ph[k] += now_ - tprev; tprev = now_; } } while (0)
  long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
  constexpr int LD = SignWave32::LD;
  const int lane = lane_id();
  const long long c0 = dbg ? (long long)__builtin_readcyclecounter() : 0;
  const int r16 = lane & 15, kk = lane >> 4;
  double* scr = S + 32 * LD;
  // S_0 = X / ||X||_F.  The Frobenius norm is the 2-norm of the svec itself (the sqrt2 on the off-diagonals counts
  // them twice), so it comes from the loaded values with one wave reduction -- no pass over the tile in LDS; like the
  // 1-norm it bounds the spectral radius, and the first step re-normalises by ||Y||_F^(1/2) anyway (sign_sched.h).
  {
    double v[16];
    sw32_load_cols<16>(in, n, lane, 0, v);
    double ss = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) ss += v[q] * v[q];
    const double nrm = sqrt(wave_sum(ss));
    const double scale = nrm > 0.0 ? 1.0 / nrm : (nrm == 0.0 ? 0.0 : nrm);   // NaN propagates (flagged at the store)
    sw32_tile_from_cols<16>(S, lane, 0, v, scale);
  }
  wave_fence();
  double f[8][2];
  SignSched sched;
  if (hint) { const int h = __builtin_amdgcn_readfirstlane(*hint); if (h > 0) sched.lift0 = h; }
  bool last = false;
  const long long c1 = dbg ? (long long)__builtin_readcyclecounter() : 0;
  tprev = c1;
  while (!last) {
    sw32_frags(S, r16, kk, f);
    if (DBG) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    SW32_STAMP(0);
    // Y = S S: the three upper sub-tiles on the matrix cores (24 MFMAs); the lower one, needed as a register operand
    // of S Y, is the transpose of Y(0,1): 4 LDS writes + 4 reads through a 16 x 17 tile instead of 8 more MFMAs (the
    // MFMA pipe is what bounds this kernel)
    sl_v4f64 y[2][2];
    y[0][0] = y[0][1] = y[1][1] = sl_v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      y[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[s][0], f[s][0], y[0][0], 0, 0, 0);
      y[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[s][0], f[s][1], y[0][1], 0, 0, 0);
      y[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[s][1], f[s][1], y[1][1], 0, 0, 0);
    }
    if (DBG) { asm volatile("s_nop 0" :: "v"(y[0][0][0]), "v"(y[0][1][0]), "v"(y[1][1][0]) : "memory"); }
    SW32_STAMP(1);
#pragma unroll
    for (int r = 0; r < 4; ++r) scr[r16 * SignWave32::SCR_LD + kk + 4 * r] = y[0][1][r];   // element (kk+4r, r16) -> scr[r16][kk+4r]
    // tr Y and ||Y||_F^2 (the off-diagonal sub-tile counts twice)
    double pa = 0.0, pb = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (kk + 4 * r == r16) pa += y[0][0][r] + y[1][1][r];
      pb += y[0][0][r] * y[0][0][r] + y[1][1][r] * y[1][1][r] + 2.0 * (y[0][1][r] * y[0][1][r]);
    }
    wave_fence();
#pragma unroll
    for (int r = 0; r < 4; ++r) y[1][0][r] = scr[(kk + 4 * r) * SignWave32::SCR_LD + r16];
    wave_fence();
    // S Y on the three upper sub-tiles (24 MFMAs, no LDS traffic), then S in accumulator layout (12 LDS reads, short
    // live range: that is what keeps the kernel at 128 VGPRs, 4 wavefronts per SIMD, without spills)
    sl_v4f64 z[3], e[3];
    SW32_STAMP(2);
    sw32_mma_regB(f, y, z);
    if (DBG) { asm volatile("s_nop 0" :: "v"(z[0][0]), "v"(z[1][0]), "v"(z[2][0]) : "memory"); }
    SW32_STAMP(3);
    const double ta = wave_sum(pa), tb = wave_sum(pb);
    sw32_dlayout(S, r16, kk, e);
    double pg = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double d0 = e[0][r] - z[0][r], d1 = e[1][r] - z[1][r], d2 = e[2][r] - z[2][r];
      pg += d0 * d0 + d2 * d2 + 2.0 * (d1 * d1);
    }
    const double tg = wave_sum(pg);
    const double mu = sched.decide<false>(n, ta, tb, tg, last);
    if (DBG) { asm volatile("s_nop 0" :: "v"(mu) : "memory"); }
    SW32_STAMP(4);
    const double alpha = -0.5 * mu * mu * mu, beta = 1.5 * mu;
    sl_v4f64 t[3];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) t[q][r] = alpha * z[q][r] + beta * e[q][r];
    wave_fence();                                                 // all reads of S are done
    sw32_store(S, r16, kk, t);
    wave_fence();
    if (DBG) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
    SW32_STAMP(5);
  }
#undef SW32_STAMP
  if (steps_out && lane == 0) *steps_out = sched.steps;
  if (hint && lane == 0) *hint = sched.lifts;
  const long long c2 = dbg ? (long long)__builtin_readcyclecounter() : 0;
  // P = 0.5 (X0 + S X0): A fragments of S from LDS, then LDS is reused for X0, whose sub-tiles are read in accumulator
  // layout (register B operand).  The lower sub-tile of X0 is read directly too (X0 is exactly symmetric in LDS).
  sw32_frags(S, r16, kk, f);
  wave_fence();
#pragma unroll 1
  for (int q0 = 0; q0 < 16; q0 += 4) {   // X0 again (L2-hot), four columns per lane at a time: the fragments f are live
    double v[4];
    sw32_load_cols<4>(in, n, lane, q0, v);
    sw32_tile_from_cols<4>(S, lane, q0, v, 1.0);
  }
  wave_fence();
  sl_v4f64 x0[3], xb[2][2], p[3];
  sw32_dlayout(S, r16, kk, x0);
  xb[0][0] = x0[0]; xb[0][1] = x0[1]; xb[1][1] = x0[2];
#pragma unroll
  for (int r = 0; r < 4; ++r) xb[1][0][r] = S[(16 + kk + 4 * r) * LD + r16];
  sw32_mma_regB(f, xb, p);
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) p[q][r] = 0.5 * p[q][r] + 0.5 * x0[q][r];
  wave_fence();
  sw32_store(S, r16, kk, p);
  wave_fence();
  const bool bad = sw32_store_cols(S, out, n, lane);
  if (bad && fail) atomicAdd(fail, 1);
  if (dbg && lane == 0) {   // developer aid (CUADMM_PSD_DEBUG): cycles of prologue / iteration / epilogue, steps
    const long long c3 = (long long)__builtin_readcyclecounter();
    dbg[0] = c1 - c0; dbg[1] = c2 - c1; dbg[2] = c3 - c2; dbg[3] = sched.steps;
    if (DBG) for (int q = 0; q < 6; ++q) dbg[4 + q] = ph[q];
  }
}

}  // namespace cuadmm
