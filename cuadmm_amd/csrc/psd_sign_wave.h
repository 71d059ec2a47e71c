// PSD projection of blocks with 9 <= n <= 64 by the matrix-sign iteration with ONE WAVEFRONT PER BLOCK (NT x NT sub-tiles of
// 16 x 16: NT = 1: n <= 16, 2: n <= 32, 3: n <= 48, 4: n <= 64), optionally FUSED with the vector work of the ADMM iteration
// on the block's svec range (psd_fuse.h) -- the kernels of BASELINE configs[1] and [3].
//
// Replaces, per block (reference src/solver.cu:534-647): vector_to_matrices, the eigendecomposition, max(W, 0), V diag,
// the DGEMM and matrices_to_vector; fused: also solver.cu:514-527 (A^T y, Rd1, Xb), :652-656,746-758,774-776 (S, Rd, X
// update, the two sums) and the block-local rows of :478,695,764 (A X, A (S - C)).
//
// Why one wavefront: the one-workgroup-per-block kernel (psd_sign_lds_kernel, 6 / 10 wavefronts, operands read from LDS
// for every MFMA) is bound by LDS bandwidth -- two fragment reads per MFMA, 35 % of the fp64 matrix-core peak on the
// n = 45 blocks that dominate configs[3].  With one wavefront owning the whole block
//   * the 4 NT x NT operand fragments of S are read from LDS ONCE per step (f[s][x] = S[4 s + kk][16 x + r16]); they are the
//     A fragments of both products (S symmetric), the B fragments of Y = S S, and -- register 4 i + r of column block j is
//     element (16 i + 4 r + kk, 16 j + r16) -- the accumulator-layout copy of S that the combine step needs;
//   * Y = S^2 stays in registers: the accumulator layout of v_mfma_f64_16x16x4_f64 is the B-operand layout, so the upper
//     sub-tiles feed S Y directly; the lower ones are transposed through LDS (the region of S is dead once the fragments
//     are loaded, so the scratch tiles alias it): 4 writes + 4 reads per off-diagonal sub-tile instead of 4 NT MFMAs;
//   * only sub-tiles on or above the diagonal are computed (1 / 3 / 6 / 10) and the next iterate is stored mirrored
//     (exactly symmetric iterate, psd_large.hip explains why that matters).
// Per step and block: 2 * NU * 4 NT MFMAs (NT = 3: 144) against 4 NT^2 + 8 NL LDS reads and <= 8 NU + 4 NL writes.
// Registers: fragments 8 NT^2, Y 8 NT^2, S Y 8 NU VGPRs -> eight / four / two / one wavefront per SIMD.  LDS: NP x (NP + 1)
// doubles per block.  A handful of 33 <= n <= 64 blocks is served by the one-workgroup kernels instead (latency; the
// planner switches by block count).
#pragma once
#include <hip/hip_runtime.h>

#include "psd_device.h"
#include "psd_fuse.h"
#include "psd_sign_lds.h"
#include "sign_sched.h"
#include "wave_reduce.h"

namespace cuadmm {

// e -> (r, c) of the column-major upper triangle (e = c (c + 1) / 2 + r, r <= c), packed r | c << 8, for every e < 2080
// (n <= 64).  The map does not depend on n, so ONE 4 KB table (L1-resident) serves every block: the svec is read and
// written FLAT -- lane l takes elements l, l + 64, ... : fully coalesced 512-byte accesses, no index decoding (tri_decode
// costs a float sqrt and two correction loops per element), and the loops over a block's svec are rolled in batches of U
// loads.  (Round 2's first version walked column pairs with everything unrolled: the prologue and epilogue of a fused
// n <= 48 block were 60 KB of straight-line code -- the instruction cache of a CU pair holds 64 KB -- and took 39 % of a
// block's lifetime; measured with CUADMM_PSD_DEBUG_GEN.)
struct SwtTab {
  unsigned short v[2080];
  constexpr SwtTab() : v() {
    int e = 0;
    for (int c = 0; c < 64; ++c)
      for (int r = 0; r <= c; ++r) v[e++] = (unsigned short)(r | (c << 8));
  }
};
__device__ const SwtTab g_swt_tab = SwtTab();

template <int NT>
struct SignWaveT {
  static constexpr int NP = 16 * NT, LD = NP + 1, KS = 4 * NT;
  static constexpr int NU = NT * (NT + 1) / 2;                 // sub-tiles on or above the diagonal
  static constexpr int MAXLEN = NP * (NP + 1) / 2;
  static constexpr int NSLOT = (MAXLEN + 63) / 64;             // svec elements per lane (NT = 1: 3, 2: 9, 3: 19, 4: 33)
  static constexpr int U = NT == 1 ? 3 : (NT == 2 ? 9 : (NT == 3 ? 10 : 11));   // loads in flight per batch
  static constexpr size_t LDS_BYTES = sizeof(double) * NP * LD;
  static constexpr int SCR_LD = 17;                            // transposition tiles (alias the region of S)
};

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
// ksteps: k-steps (of 4 rows / columns) that hold at least one REAL row of the block, (n + 3) / 4 -- the steps beyond are exact
// zeros in both operands (the padding of the tile) and are skipped: a wave-uniform scalar branch instead of 64 matrix-core cycles
// (n = 28 on the 32 x 32 tile: one step in eight; n = 10 on 16 x 16: one in four; the tiny blocks of a closed plan: up to three)
template <int NT, int B0, int B1>
__device__ __forceinline__ void swt_mma_regB(const double (&fa)[4 * NT][NT], const sl_v4f64 (&yb)[NT][NT], sl_v4f64 (&acc)[NT][NT], int ksteps = 4 * NT) {
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
      if (4 * b + s < ksteps) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = i; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[4 * b + s][i], yb[b][j][s], acc[i][j], 0, 0, 0);
      }
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

// One batch of the flat svec walk: slot u of the batch is element e = base + 64 u + lane.  An invalid slot (e >= len) reads
// element 0 of the block and is masked afterwards, so the loads of a batch are straight-line: one memory round trip per batch.
// developer aid (fused kernels, CUADMM_CU_DBG): tick stamp k after draining the memory counters, so that a phase owns its waits
#define CUADMM_SWT_STAMP(k)                                                                                   \
  if (FUSED && dbg) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (lane == 0) dbg[k] = (long long)__builtin_readcyclecounter() - c0; }
#define CUADMM_SWT_SLOT(u)                                        \
  const int e_ = base + 64 * (u) + lane;                          \
  const bool ok_ = e_ < len;                                      \
  const int ec_ = ok_ ? e_ : 0

// FUSED: `in` / `out` are unused; fz carries the vectors, off = svec offset of the block, slot = its partial-sum slot,
// id = its index in the plan (local constraint rows), poff = offset of this iteration's partial arrays (several iterations
// per launch: SignFuse::iters)
// MEGA: the schedule with the mega-lift (sign_sched.h) -- everywhere except the 64-register instantiation (n <= 16 at eight wavefronts
// per SIMD: a class that fills the chip), whose budget its state does not fit
template <int NT, bool FUSED, bool MEGA = true>
__device__ __forceinline__ void psd_sign_wave_body(const double* __restrict__ in, double* __restrict__ out, int n, int* fail, double* S,
                                                   int* steps_out, int* hint, long long* dbg, const SignFuse& fz, long long off, int slot, int id,
                                                   long long poff = 0) {
  using Cfg = SignWaveT<NT>;
  constexpr int LD = Cfg::LD, NP = Cfg::NP, U = Cfg::U;
  const int lane = lane_id();
  const int r16 = lane & 15, kk = lane >> 4;
  const int len = n * (n + 1) / 2;
  const int ksteps = (n + 3) >> 2;                        // k-steps with at least one real row (the rest of the tile is exact zeros)
  const unsigned short* __restrict__ tab = g_swt_tab.v;
  const long long c0 = dbg ? (long long)__builtin_readcyclecounter() : 0;
  // ---- prologue: the tile takes X (fused: Xb = X + sigma (A^T y - C), formed here); ||X||_F is the 2-norm of the svec (the
  // sqrt2 counts the off-diagonals twice) and goes onto the fragments of the first step
  if (n < NP) {                          // a full-size block writes every entry of the tile itself
#pragma unroll 1
    for (int e = lane; e < NP * LD; e += 64) S[e] = 0.0;
  }
  if (dbg && !FUSED) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (lane == 0) dbg[4] = (long long)__builtin_readcyclecounter() - c0; }
  const bool local_rows = FUSED && fz.lc != nullptr;      // closed blocks have their own body (psd_sign_closed.h)
  LcDesc lcd = {0, 0, 0, 0};
  if (local_rows) lcd = fz.lc[id];       // {first row, rows | longest row << 16, first nonzero, nonzeros}: wave-uniform
  double ss = 0.0;
#pragma unroll 1
  for (int base = 0; base < len; base += 64 * U) {
    double v[U];
    int rc[U];
    if (FUSED) {
      int p0[U], p1[U];
      double cq[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        CUADMM_SWT_SLOT(u);
        const long long i = off + ec_;
        rc[u] = tab[ec_];
        p0[u] = fz.rp[i]; p1[u] = ok_ ? fz.rp[i + 1] : p0[u]; cq[u] = fz.C[i]; v[u] = fz.X[i];
      }
      if (base == 0) { CUADMM_SWT_STAMP(5); }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        CUADMM_SWT_SLOT(u);
        double t = 0.0;
        for (int p = p0[u]; p < p1[u]; ++p) t += fz.av[p] * fz.y[fz.ci[p]];
        const double r1 = t - cq[u];
        if (ok_) fz.Rd1[off + e_] = r1;
        v[u] = ok_ ? v[u] + r1 * fz.sig : 0.0;
        (void)ec_;
      }
      if (base == 0) { CUADMM_SWT_STAMP(6); }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        CUADMM_SWT_SLOT(u);
        rc[u] = tab[ec_];
        const double x = in[ec_];
        v[u] = ok_ ? x : 0.0;
      }
    }
    if (dbg && !FUSED && base == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (lane == 0) dbg[5] = (long long)__builtin_readcyclecounter() - c0; }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      CUADMM_SWT_SLOT(u);
      (void)ec_;
      ss += v[u] * v[u];
      const int r = rc[u] & 255, c = rc[u] >> 8;
      const double x = (r == c) ? v[u] : v[u] * kSqrt2Inv;
      if (ok_) { S[r * LD + c] = x; S[c * LD + r] = x; }
    }
  }
  if (dbg && !FUSED) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (lane == 0) dbg[6] = (long long)__builtin_readcyclecounter() - c0; }
  const double nrm = sqrt(wave_sum(ss));
  const double scale = nrm > 0.0 ? 1.0 / nrm : (nrm == 0.0 ? 0.0 : nrm);   // NaN propagates (flagged at the store)
  wave_fence();
  double f[4 * NT][NT];
  SignSchedT<MEGA> sched;
  if (hint) { const int h = __builtin_amdgcn_readfirstlane(*hint); if (h > 0) sched.lift0 = h; }
  bool last = false;
  const long long c1 = dbg ? (long long)__builtin_readcyclecounter() : 0;
  while (!last) {
    swt_frags<NT>(S, r16, kk, f);
    if (sched.steps == 0) {                                // S_0 = X / ||X||_F: the tile holds X, the scale goes onto its fragments
#pragma unroll
      for (int s = 0; s < 4 * NT; ++s)
#pragma unroll
        for (int x = 0; x < NT; ++x) f[s][x] *= scale;
    }
    wave_fence();                                          // the region of S is scratch from here to the store of the next iterate
    // Y = S S on the upper sub-tiles
    sl_v4f64 y[NT][NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = i; j < NT; ++j) y[i][j] = sl_v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4 * NT; ++s)
      if (s < ksteps) {                                     // k-steps of pure padding are skipped (swt_mma_regB)
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
          for (int j = i; j < NT; ++j) y[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[s][i], f[s][j], y[i][j], 0, 0, 0);
      }
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
    swt_mma_regB<NT, 0, 1>(f, y, z, ksteps);                        // row block 0 of Y: upper sub-tiles only
    swt_lower_read<NT>(S, r16, kk, y);
    swt_mma_regB<NT, 1, NT>(f, y, z, ksteps);
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
      mu = sched.template decide<false>(n, ta, tb, tg, last);
    } else {
      mu = sched.template decide<false>(n, 0.0, 0.0, 0.0, last);    // a branch that does not read them
    }
    double alpha, beta;
    sched.coefs(mu, alpha, beta);
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = i; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) z[i][j][r] = fma(alpha, z[i][j][r], beta * f[4 * i + r][j]);   // one contraction, spelled out: every instantiation rounds alike
    wave_fence();                                            // the scratch tiles have been read
    swt_store_mirrored<NT>(S, r16, kk, z);
    wave_fence();
  }
  if (steps_out && lane == 0) *steps_out = sched.steps;
  if (hint && lane == 0) *hint = sched.lifts;
  const long long c2 = dbg ? (long long)__builtin_readcyclecounter() : 0;
  // ---- epilogue: P = 0.5 (X0 + S X0).  A fragments of S from LDS, then the tile takes X0 again (fused: rebuilt from X and
  // Rd1, bit-identical), whose sub-tiles are read in accumulator layout (register B operand; X0 is exactly symmetric in LDS)
  swt_frags<NT>(S, r16, kk, f);
  wave_fence();
#pragma unroll 1
  for (int base = 0; base < len; base += 64 * U) {
    double v[U];
    int rc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      CUADMM_SWT_SLOT(u);
      (void)ok_;
      rc[u] = tab[ec_];
      v[u] = FUSED ? fz.X[off + ec_] + fz.Rd1[off + ec_] * fz.sig : in[ec_];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      CUADMM_SWT_SLOT(u);
      (void)ec_;
      const int r = rc[u] & 255, c = rc[u] >> 8;
      const double x = (r == c) ? v[u] : v[u] * kSqrt2Inv;
      if (ok_) { S[r * LD + c] = x; S[c * LD + r] = x; }     // same positions as in the prologue: the zero padding is still there
    }
  }
  wave_fence();
  CUADMM_SWT_STAMP(7);
  {
    sl_v4f64 xb[NT][NT], p[NT][NT];
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) xb[b][j][r] = S[(16 * b + kk + 4 * r) * LD + 16 * j + r16];
    swt_mma_regB<NT, 0, NT>(f, xb, p, ksteps);
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = i; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) p[i][j][r] = 0.5 * p[i][j][r] + 0.5 * xb[i][j][r];
    wave_fence();
    swt_store_mirrored<NT>(S, r16, kk, p);
  }
  wave_fence();
  CUADMM_SWT_STAMP(8);
  // ---- the projection leaves through the flat walk again.  Fused: S, Rd, X updates and the two sums on the way (the
  // expressions of post_kernel); the slot of P(r, c) in the tile then takes S - C and its mirror image (the pad column for a
  // diagonal element) the new X -- the staging the local constraint rows read.
  bool bad = false;
  double s_rd = 0.0, s_cx = 0.0;
  const bool upd = FUSED && fz.mode == 0;
  int ze = 0, kb = 0, ke = 0, krow = 0;
  double zv = 0.0;
  if (local_rows) {                      // index data of the local rows first: the round trip overlaps the walk below
    if (lane < lcd.w) { ze = fz.lc_e[lcd.z + lane]; zv = fz.lc_v[lcd.z + lane]; }
    if (lane < (lcd.y & 0xffff)) { kb = fz.lc_nzptr[lcd.x + lane] - lcd.z; ke = fz.lc_nzptr[lcd.x + lane + 1] - lcd.z; krow = fz.lc_row[lcd.x + lane]; }
  }
#pragma unroll 1
  for (int base = 0; base < len; base += 64 * U) {
    double pq[U], xq[U], rq[U], cq[U];
    int rc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      CUADMM_SWT_SLOT(u);
      (void)ok_;
      rc[u] = tab[ec_];
      if (FUSED) {
        xq[u] = fz.X[off + ec_]; rq[u] = fz.Rd1[off + ec_];
        cq[u] = (upd || local_rows) ? fz.C[off + ec_] : 0.0;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) pq[u] = S[(rc[u] & 255) * LD + (rc[u] >> 8)];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      CUADMM_SWT_SLOT(u);
      (void)ec_;
      const int r = rc[u] & 255, c = rc[u] >> 8;
      const double pm = pq[u];
      bad |= ok_ && !(fabs(pm) <= 1.7976931348623157e308);
      const double xp = (r == c) ? pm : pm * kSqrt2;         // Xproj[e]
      if (FUSED) {
        const long long i = off + e_;
        const double x = xq[u], r1 = rq[u];
        const double xdiff = xp - x;
        const double sv = fz.inv_sig * xdiff - r1;
        if (ok_) fz.S[i] = sv;
        double xn = x;
        if (upd) {
          const double rd = r1 + sv;
          xn = x + fz.tau_sig * rd;
          if (ok_) fz.X[i] = xn;
          s_rd += ok_ ? rd * rd : 0.0;
          s_cx += ok_ ? cq[u] * xn : 0.0;
        }
        if (local_rows && ok_) {
          S[r * LD + c] = sv - cq[u];
          S[r == c ? r * LD + NP : c * LD + r] = xn;
        }
      } else {
        if (ok_) out[e_] = xp;
      }
    }
  }
  CUADMM_SWT_STAMP(9);
  if (FUSED) {
    if (local_rows && lcd.w > 0) {
      // one lane per nonzero forms a * v from the staging, one lane per row adds its segment in order (host: a block keeps its
      // local rows only when it has at most 64 of them with at most 64 nonzeros in total)
      wave_fence();
      const int zt = tab[ze];
      const int zr = zt & 255, zc = zt >> 8;
      const double ps = zv * S[zr * LD + zc];
      const double px = zv * S[zr == zc ? zr * LD + NP : zc * LD + zr];
      const int maxlen = lcd.y >> 16;
      double as = 0.0, ax = 0.0;
      for (int t = 0; t < maxlen; ++t) {
        const int src = (kb + t) & 63;
        const double vs = __shfl(ps, src, 64), vx = __shfl(px, src, 64);
        if (kb + t < ke) { as += vs; ax += vx; }
      }
      if (lane < (lcd.y & 0xffff)) {
        if (fz.outS) fz.outS[krow] = as;
        if (upd && fz.outX) fz.outX[krow] = ax;
      }
    }
    if (upd) {
      s_rd = wave_sum(s_rd);
      s_cx = wave_sum(s_cx);
      if (lane == 0) { fz.partials[poff + 2 * (long long)slot] = s_rd; fz.partials[poff + 2 * (long long)slot + 1] = s_cx; }
    }
  }
  if (bad && fail) atomicAdd(fail, 1);
  if (dbg && lane == 0) {   // developer aid (CUADMM_PSD_DEBUG): cycles of prologue / iteration / epilogue, steps
    const long long c3 = (long long)__builtin_readcyclecounter();
    dbg[0] = c1 - c0; dbg[1] = c2 - c1; dbg[2] = c3 - c2; dbg[3] = sched.steps;
  }
}
#undef CUADMM_SWT_SLOT
#undef CUADMM_SWT_STAMP

}  // namespace cuadmm
