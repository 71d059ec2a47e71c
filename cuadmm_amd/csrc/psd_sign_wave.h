// PSD projection of blocks with 32 < n <= 64 by the matrix-sign iteration with ONE WAVEFRONT PER BLOCK (NT = 3: n <= 48,
// NT = 4: n <= 64) -- the generalisation of the n <= 32 kernel of psd_sign_lds.h (SignWave32).
//
// Replaces, per block (reference src/solver.cu:534-647): vector_to_matrices, the eigendecomposition, max(W, 0), V diag,
// the DGEMM and matrices_to_vector.
//
// Why: the one-workgroup-per-block kernel (psd_sign_lds_kernel, 6 / 10 wavefronts, operands read from LDS for every
// MFMA) is bound by LDS bandwidth -- two fragment reads per MFMA, 35 % of the fp64 matrix-core peak on the n = 45
// blocks that dominate BASELINE configs[3].  With one wavefront owning the whole block
//   * the 4 NT x NT operand fragments of S are read from LDS ONCE per step (f[s][x] = S[4 s + kk][16 x + r16]); they are the
//     A fragments of both products (S symmetric), the B fragments of Y = S S, and -- register 4 i + r of column block j is
//     element (16 i + 4 r + kk, 16 j + r16) -- the accumulator-layout copy of S that the combine step needs;
//   * Y = S^2 stays in registers: the accumulator layout of v_mfma_f64_16x16x4_f64 is the B-operand layout, so the upper
//     sub-tiles feed S Y directly; the lower ones are transposed through LDS (the region of S is dead once the fragments
//     are loaded, so the scratch tiles alias it): 4 writes + 4 reads per off-diagonal sub-tile instead of 4 NT MFMAs;
//   * only sub-tiles on or above the diagonal are computed (6 of 9 / 10 of 16) and the next iterate is stored mirrored
//     (exactly symmetric iterate, psd_large.hip explains why that matters).
// Per step and block: 2 * NU * 4 NT MFMAs (NT = 3: 144) against 4 NT^2 + 8 NL LDS reads and <= 8 NU + 4 NL writes --
// the matrix cores are the only busy unit.  Registers: fragments 8 NT^2, Y 8 NT^2, S Y 8 NU VGPRs (NT = 3: 72 + 72 + 48
// -> two wavefronts per SIMD; NT = 4: 128 + 128 + 80 -> one).  LDS: NP x (NP + 1) doubles per block (18.4 / 32.5 KB).
#pragma once
#include <hip/hip_runtime.h>

#include "psd_device.h"
#include "psd_fuse.h"
#include "psd_sign_lds.h"
#include "sign_sched.h"
#include "wave_reduce.h"

namespace cuadmm {

template <int NT>
struct SignWaveT {
  static constexpr int NP = 16 * NT, LD = NP + 1, KS = 4 * NT;
  static constexpr int NU = NT * (NT + 1) / 2;                 // sub-tiles on or above the diagonal
  // svec <-> lanes by COLUMN PAIRS (swt_slot): W lanes serve one pair, G pairs per load
  static constexpr int W = NT == 1 ? 16 : (NT == 2 ? 32 : 64), G = 64 / W, MC = W - 1;
  static constexpr bool EXTRA = NP > MC;                       // n = W: column W - 1 takes one more load
  static constexpr int NPAIRS = (NP / 2 < (MC + 1) / 2 ? NP / 2 : (MC + 1) / 2);
  static constexpr int NQ = (NPAIRS + G - 1) / G + (EXTRA ? 1 : 0);   // global loads per lane (NT = 1: 3, 2: 9, 3: 24, 4: 33)
  static constexpr size_t LDS_BYTES = sizeof(double) * NP * LD;
  static constexpr int SCR_LD = 17;                            // transposition tiles (alias the region of S)
};

// svec <-> lanes: column c of the upper triangle is the contiguous svec range [c (c + 1) / 2, + c + 1).  A group of W lanes
// takes the column pair (p, m - 1 - p), m = min(n, W - 1): p + 1 + m - p = m + 1 <= W lanes, two coalesced runs, no index
// decoding (tri_decode costs a float sqrt and two correction loops per element).  n = W: column W - 1 is one more load.
// Returns whether the lane holds an element.
template <int NT>
__device__ __forceinline__ bool swt_slot(int q, int lane, int n, int& r, int& c) {
  using Cfg = SignWaveT<NT>;
  if (Cfg::EXTRA && q == Cfg::NQ - 1) { r = lane; c = Cfg::MC; return n > Cfg::MC && lane <= Cfg::MC; }
  const int g = lane / Cfg::W, l = lane % Cfg::W;
  const int m = n < Cfg::MC ? n : Cfg::MC;
  const int c1 = q * Cfg::G + g, c2 = m - 1 - c1;
  if (c1 > c2) return false;
  if (l <= c1) { r = l; c = c1; return true; }
  r = l - c1 - 1; c = c2;
  return c2 != c1 && r <= c2;
}

template <int NT>
__device__ __forceinline__ void swt_load(const double* __restrict__ src, int n, int lane, double (&v)[SignWaveT<NT>::NQ]) {
#pragma unroll
  for (int q = 0; q < SignWaveT<NT>::NQ; ++q) {
    int r, c;
    v[q] = swt_slot<NT>(q, lane, n, r, c) ? src[c * (c + 1) / 2 + r] : 0.0;
  }
}
// M = scale * smat(v), both triangles.  M must have been zeroed where no element lands (padding rows / columns >= n).
template <int NT>
__device__ __forceinline__ void swt_tile_from(double* __restrict__ M, int n, int lane, const double (&v)[SignWaveT<NT>::NQ], double scale) {
  constexpr int LD = SignWaveT<NT>::LD;
#pragma unroll
  for (int q = 0; q < SignWaveT<NT>::NQ; ++q) {
    int r, c;
    if (swt_slot<NT>(q, lane, n, r, c)) {
      const double x = v[q] * (r == c ? scale : scale * kSqrt2Inv);
      M[r * LD + c] = x;
      M[c * LD + r] = x;
    }
  }
}
template <int NT>
__device__ __forceinline__ bool swt_store_svec(const double* __restrict__ M, double* __restrict__ out, int n, int lane) {
  constexpr int LD = SignWaveT<NT>::LD;
  bool bad = false;
#pragma unroll
  for (int q = 0; q < SignWaveT<NT>::NQ; ++q) {
    int r, c;
    if (swt_slot<NT>(q, lane, n, r, c)) {
      const double x = M[r * LD + c];
      bad |= !(fabs(x) <= 1.7976931348623157e308);
      out[c * (c + 1) / 2 + r] = (r == c) ? x : x * kSqrt2;
    }
  }
  return bad;
}

// all operand fragments of the symmetric matrix in LDS: f[s][x] = M[4 s + kk][16 x + r16]
template <int NT>
__device__ __forceinline__ void swt_frags(const double* __restrict__ M, int r16, int kk, double (&f)[4 * NT][NT]) {
  constexpr int LD = SignWaveT<NT>::LD;
#pragma unroll
  for (int s = 0; s < 4 * NT; ++s)
#pragma unroll
    for (int x = 0; x < NT; ++x) f[s][x] = M[(4 * s + kk) * LD + 16 * x + r16];
}

// lower sub-tiles of a symmetric matrix whose upper sub-tiles sit in accumulator layout: t[b][j] (b > j) = t[j][b]^T through
// NT (NT - 1) / 2 scratch tiles of 16 x 17 doubles -- in two halves, so that the products that need no lower sub-tile run
// while the transposed tiles are on their way through LDS
template <int NT>
__device__ __forceinline__ void swt_lower_write(double* __restrict__ scr, int r16, int kk, const sl_v4f64 (&t)[NT][NT]) {
  constexpr int SL = SignWaveT<NT>::SCR_LD;
  int slot = 0;
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = i + 1; j < NT; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) scr[slot * 16 * SL + r16 * SL + kk + 4 * r] = t[i][j][r];   // element (kk + 4 r, r16) -> scr[r16][kk + 4 r]
      ++slot;
    }
  wave_fence();
}
template <int NT>
__device__ __forceinline__ void swt_lower_read(const double* __restrict__ scr, int r16, int kk, sl_v4f64 (&t)[NT][NT]) {
  constexpr int SL = SignWaveT<NT>::SCR_LD;
  int slot = 0;
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = i + 1; j < NT; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) t[j][i][r] = scr[slot * 16 * SL + (kk + 4 * r) * SL + r16];
      ++slot;
    }
}

// acc(upper sub-tiles) (+)= A * B over the row blocks [B0, B1) of B: A symmetric, given by its fragments; B by its sub-tiles
// in accumulator layout.  Row block 0 of B needs no lower sub-tile.
template <int NT, int B0, int B1>
__device__ __forceinline__ void swt_mma_regB(const double (&fa)[4 * NT][NT], const sl_v4f64 (&yb)[NT][NT], sl_v4f64 (&acc)[NT][NT]) {
  if (B0 == 0) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = i; j < NT; ++j) acc[i][j] = sl_v4f64{0.0, 0.0, 0.0, 0.0};
  }
#pragma unroll
  for (int b = B0; b < B1; ++b)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = i; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[4 * b + s][i], yb[b][j][s], acc[i][j], 0, 0, 0);
}

// mirrored store of the upper sub-tiles (on a diagonal sub-tile the upper triangle decides)
template <int NT>
__device__ __forceinline__ void swt_store_mirrored(double* __restrict__ M, int r16, int kk, const sl_v4f64 (&d)[NT][NT]) {
  constexpr int LD = SignWaveT<NT>::LD;
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = i; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * i + kk + 4 * r, col = 16 * j + r16;
        if (i != j || col >= row) { M[row * LD + col] = d[i][j][r]; M[col * LD + row] = d[i][j][r]; }
      }
}

// FUSED: `in` / `out` are unused; fz carries the vectors, off = svec offset of the block, slot = its partial-sum slot
template <int NT, bool FUSED>
__device__ __forceinline__ void psd_sign_wave_body(const double* __restrict__ in, double* __restrict__ out, int n, int* fail, double* S,
                                                   int* steps_out, int* hint, long long* dbg, const SignFuse& fz, long long off, int slot) {
  using Cfg = SignWaveT<NT>;
  constexpr int LD = Cfg::LD, NP = Cfg::NP;
  const int lane = lane_id();
  const int r16 = lane & 15, kk = lane >> 4;
  const long long c0 = dbg ? (long long)__builtin_readcyclecounter() : 0;
  // S_0 = X / ||X||_F: the Frobenius norm is the 2-norm of the svec (the sqrt2 counts the off-diagonals twice)
  {
    double v[Cfg::NQ];
    if (FUSED) {
      // Straight-line loads (an invalid slot reads the block's first element and is masked afterwards): all row pointers /
      // C / X of the lane's slots are in flight before the first use; then the (mostly empty) rows of A^T.
      int p0[Cfg::NQ], p1[Cfg::NQ];
      double cq[Cfg::NQ];
#pragma unroll
      for (int q = 0; q < Cfg::NQ; ++q) {
        int r, c;
        const bool ok = swt_slot<NT>(q, lane, n, r, c);
        const long long i = off + (ok ? c * (c + 1) / 2 + r : 0);
        p0[q] = fz.rp[i]; p1[q] = fz.rp[i + 1]; cq[q] = fz.C[i]; v[q] = fz.X[i];
      }
#pragma unroll
      for (int q = 0; q < Cfg::NQ; ++q) {
        int r, c;
        const bool ok = swt_slot<NT>(q, lane, n, r, c);
        double t = 0.0;
        for (int p = p0[q]; p < (ok ? p1[q] : p0[q]); ++p) t += fz.av[p] * fz.y[fz.ci[p]];
        const double r1 = t - cq[q];
        if (ok) fz.Rd1[off + c * (c + 1) / 2 + r] = r1;
        v[q] = ok ? v[q] + r1 * fz.sig : 0.0;
      }
    } else {
      swt_load<NT>(in, n, lane, v);
    }
    for (int e = lane; e < NP * LD; e += 64) S[e] = 0.0;
    double ss = 0.0;
#pragma unroll
    for (int q = 0; q < Cfg::NQ; ++q) ss += v[q] * v[q];
    const double nrm = sqrt(wave_sum(ss));
    const double scale = nrm > 0.0 ? 1.0 / nrm : (nrm == 0.0 ? 0.0 : nrm);   // NaN propagates (flagged at the store)
    wave_fence();
    swt_tile_from<NT>(S, n, lane, v, scale);
  }
  wave_fence();
  double f[4 * NT][NT];
  SignSched sched;
  if (hint) { const int h = __builtin_amdgcn_readfirstlane(*hint); if (h > 0) sched.lift0 = h; }
  bool last = false;
  const long long c1 = dbg ? (long long)__builtin_readcyclecounter() : 0;
  while (!last) {
    swt_frags<NT>(S, r16, kk, f);
    wave_fence();                                          // the region of S is scratch from here to the store of the next iterate
    // Y = S S on the upper sub-tiles
    sl_v4f64 y[NT][NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = i; j < NT; ++j) y[i][j] = sl_v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4 * NT; ++s)
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = i; j < NT; ++j) y[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[s][i], f[s][j], y[i][j], 0, 0, 0);
    // tr Y and ||Y||_F^2 (off-diagonal sub-tiles count twice) -- only when the schedule will look at them
    const bool stats = sched.needs_stats();                 // wave-uniform
    double pa = 0.0, pb = 0.0;
    if (stats) {
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = i; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (i == j && kk + 4 * r == r16) pa += y[i][j][r];
            pb += (i == j ? 1.0 : 2.0) * (y[i][j][r] * y[i][j][r]);
          }
    }
    sl_v4f64 z[NT][NT];
    swt_lower_write<NT>(S, r16, kk, y);
    swt_mma_regB<NT, 0, 1>(f, y, z);                        // row block 0 of Y: upper sub-tiles only
    swt_lower_read<NT>(S, r16, kk, y);
    swt_mma_regB<NT, 1, NT>(f, y, z);
    double mu;
    if (stats) {
      const double ta = wave_sum(pa), tb = wave_sum(pb);
      double pg = 0.0;
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = i; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double d = f[4 * i + r][j] - z[i][j][r];     // f[4 i + r][j] = S(16 i + 4 r + kk, 16 j + r16): accumulator layout
            pg += (i == j ? 1.0 : 2.0) * (d * d);
          }
      const double tg = wave_sum(pg);
      mu = sched.decide<false>(n, ta, tb, tg, last);
    } else {
      mu = sched.decide<false>(n, 0.0, 0.0, 0.0, last);    // a branch that does not read them
    }
    const double alpha = -0.5 * mu * mu * mu, beta = 1.5 * mu;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = i; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) z[i][j][r] = alpha * z[i][j][r] + beta * f[4 * i + r][j];
    wave_fence();                                            // the scratch tiles have been read
    swt_store_mirrored<NT>(S, r16, kk, z);
    wave_fence();
  }
  if (steps_out && lane == 0) *steps_out = sched.steps;
  if (hint && lane == 0) *hint = sched.lifts;
  const long long c2 = dbg ? (long long)__builtin_readcyclecounter() : 0;
  // P = 0.5 (X0 + S X0): A fragments of S from LDS, then LDS takes X0 again (L2-hot), whose sub-tiles are read in
  // accumulator layout (register B operand; X0 is exactly symmetric in LDS, so the lower sub-tiles are read directly)
  {
    double v[Cfg::NQ];
    if (FUSED) {                                             // Xb again from X and Rd1 (bit-identical to the prologue's value)
#pragma unroll
      for (int q = 0; q < Cfg::NQ; ++q) {
        int r, c;
        const bool ok = swt_slot<NT>(q, lane, n, r, c);
        const long long i = off + (ok ? c * (c + 1) / 2 + r : 0);
        const double xb = fz.X[i] + fz.Rd1[i] * fz.sig;
        v[q] = ok ? xb : 0.0;
      }
    } else {
      swt_load<NT>(in, n, lane, v);                          // issued first: the latency overlaps the fragment reads
    }
    swt_frags<NT>(S, r16, kk, f);
    wave_fence();
    swt_tile_from<NT>(S, n, lane, v, 1.0);                   // same positions as in the prologue: the zero padding is still there
  }
  wave_fence();
  sl_v4f64 xb[NT][NT], p[NT][NT];
#pragma unroll
  for (int b = 0; b < NT; ++b)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) xb[b][j][r] = S[(16 * b + kk + 4 * r) * LD + 16 * j + r16];
  swt_mma_regB<NT, 0, NT>(f, xb, p);
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = i; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) p[i][j][r] = 0.5 * p[i][j][r] + 0.5 * xb[i][j][r];
  wave_fence();
  swt_store_mirrored<NT>(S, r16, kk, p);
  wave_fence();
  bool bad = false;
  if (FUSED) {
    double s_rd = 0.0, s_cx = 0.0;
    const bool upd = fz.mode == 0, local_rows = fz.lc_ptr != nullptr;
    // local constraint rows: their index data is requested first (one lane per nonzero, one per row; more than 64 of either
    // are fetched later), so that the round trip overlaps the slot loads below
    int k0 = 0, k1 = 0, z0 = 0, z1 = 0, ze = 0, kb = 0, ke = 0, krow = 0;
    double zv = 0.0;
    if (local_rows) {
      k0 = fz.lc_ptr[slot]; k1 = fz.lc_ptr[slot + 1];                 // wave-uniform
      if (k0 < k1) {
        z0 = fz.lc_nzptr[k0]; z1 = fz.lc_nzptr[k1];
        if (z0 + lane < z1) { ze = fz.lc_e[z0 + lane]; zv = fz.lc_v[z0 + lane]; }
        if (k0 + lane < k1) { kb = fz.lc_nzptr[k0 + lane]; ke = fz.lc_nzptr[k0 + lane + 1]; krow = fz.lc_row[k0 + lane]; }
      }
    }
    // straight-line loads again (masked afterwards), so that the slots' X / Rd1 / C are all in flight together
    double xq[Cfg::NQ], rq[Cfg::NQ], cq[Cfg::NQ], pq[Cfg::NQ];
#pragma unroll
    for (int q = 0; q < Cfg::NQ; ++q) {
      int r, c;
      const bool ok = swt_slot<NT>(q, lane, n, r, c);
      const long long i = off + (ok ? c * (c + 1) / 2 + r : 0);
      xq[q] = fz.X[i]; rq[q] = fz.Rd1[i];
      cq[q] = (upd || local_rows) ? fz.C[i] : 0.0;
      pq[q] = S[ok ? r * LD + c : 0];
    }
#pragma unroll
    for (int q = 0; q < Cfg::NQ; ++q) {
      int r, c;
      const bool ok = swt_slot<NT>(q, lane, n, r, c);
      const long long i = off + (ok ? c * (c + 1) / 2 + r : 0);
      const double pm = pq[q];
      bad |= ok && !(fabs(pm) <= 1.7976931348623157e308);
      const double xp = (r == c) ? pm : pm * kSqrt2;         // Xproj[i]
      const double x = xq[q], r1 = rq[q];
      const double xdiff = xp - x;
      const double sv = fz.inv_sig * xdiff - r1;
      if (ok) fz.S[i] = sv;
      rq[q] = sv - cq[q];                                    // S - C (local rows)
      if (upd) {
        const double rd = r1 + sv;
        const double xn = x + fz.tau_sig * rd;
        if (ok) fz.X[i] = xn;
        xq[q] = xn;
        s_rd += ok ? rd * rd : 0.0;
        s_cx += ok ? cq[q] * xn : 0.0;
      }
    }
    if (local_rows && k0 < k1) {
      // the tile region is free now: the block's svec (new X, then S - C) in packed order, the products behind it
      const int len = n * (n + 1) / 2;
      double* __restrict__ PR = S + len;
#pragma unroll 1
      for (int pass = (upd && fz.outX) ? 0 : 1; pass < 2; ++pass) {
        double* __restrict__ dst = pass == 0 ? fz.outX : fz.outS;
        if (!dst) continue;
        wave_fence();
#pragma unroll
        for (int q = 0; q < Cfg::NQ; ++q) {
          int r, c;
          if (swt_slot<NT>(q, lane, n, r, c)) S[c * (c + 1) / 2 + r] = pass == 0 ? xq[q] : rq[q];
        }
        wave_fence();
        if (z0 + lane < z1) PR[lane] = zv * S[ze];
        for (int z = z0 + 64 + lane; z < z1; z += 64) PR[z - z0] = fz.lc_v[z] * S[fz.lc_e[z]];
        wave_fence();
        if (k0 + lane < k1) {
          double acc = 0.0;
          for (int p = kb; p < ke; ++p) acc += PR[p - z0];
          dst[krow] = acc;
        }
        for (int k = k0 + 64 + lane; k < k1; k += 64) {
          double acc = 0.0;
          for (int p = fz.lc_nzptr[k]; p < fz.lc_nzptr[k + 1]; ++p) acc += PR[p - z0];
          dst[fz.lc_row[k]] = acc;
        }
      }
    }
    if (fz.mode == 0) {
      s_rd = wave_sum(s_rd);
      s_cx = wave_sum(s_cx);
      if (lane == 0) { fz.partials[2 * (long long)slot] = s_rd; fz.partials[2 * (long long)slot + 1] = s_cx; }
    }
  } else {
    bad = swt_store_svec<NT>(S, out, n, lane);
  }
  if (bad && fail) atomicAdd(fail, 1);
  if (dbg && lane == 0) {   // developer aid (CUADMM_PSD_DEBUG): cycles of prologue / iteration / epilogue, steps
    const long long c3 = (long long)__builtin_readcyclecounter();
    dbg[0] = c1 - c0; dbg[1] = c2 - c1; dbg[2] = c3 - c2; dbg[3] = sched.steps;
  }
}

}  // namespace cuadmm
