// REJECTED EXPERIMENT (round 4) -- not part of the library, not compiled by cuadmm_amd/build.py.  Kept as the record of the
// "two wavefronts per block for 33 <= n <= 48" kernel that VERDICT r3 asked for: correct (every bit-identity test of
// tests/test_gpu_batch.py and tests/test_gpu_fused.py passed on it), and SLOWER: BASELINE configs[3] 2.32 ms per iteration against
// 1.59 ms with the one-wavefront kernel (433 vs 630 iters/s).  Why: on gfx950 the vector ALU and the fp64
// MFMAs issue on the same port (tools/ubench/mfma_coissue.hip); this kernel issues 911 VALU instructions per step and wavefront
// for 72 MFMAs (the schedule's state machine runs on both wavefronts, Y travels through LDS, diagonal tiles are masked per store)
// where the one-wavefront kernel issues ~560 for 144, plus 346 scratch instructions at the 128-VGPR budget and 4-5 workgroup
// barriers per step.  More wavefronts do not help a kernel whose bound is issue cycles.
// The whole ADMM iteration of a CLOSED block with 33 <= n <= 48 on TWO wavefronts (one workgroup of 128 threads per block).
//
// Why: the one-wavefront kernel of this size (psd_sign_closed.h, NT = 3) needs 245 VGPRs and 18.8 KB of LDS per block: two
// wavefronts per SIMD.  A block in its prologue or epilogue then leaves ONE wavefront on its SIMD, whose dependent chain (LDS
// round trips, statistics, the state machine) keeps the fp64 matrix pipe at about half -- measured 1.58 ms per 16 667 blocks of
// n = 45 (BASELINE configs[3]) against 0.75 ms of MFMA issue.  Here the six upper sub-tiles of the iterate are split 3 | 3:
//     wavefront 0: (0,0) (0,1) (0,2)        wavefront 1: (1,1) (1,2) (2,2)
// each wavefront holds the operand fragments its tiles need (36 / 24 doubles) and three accumulator tiles: <= 128 VGPRs, so FOUR
// wavefronts of four different blocks share a SIMD at the same 18.8 KB of LDS per block (eight blocks per CU), and a block's
// critical path per step is 72 MFMAs instead of 144.  Y = S S leaves the registers: every tile of Y goes to the tile's upper
// storage (the iterate's fragments are in registers by then) and the second product reads its B operands from there.
//
// Same arithmetic as the one-wavefront kernel, in the same order: per output tile the k-steps run 0 .. 11; the per-lane partial
// sums of the statistics (and of ||Xb||_F^2, sum Rd^2, <C, X>) run over the slots / tiles in the one-wavefront kernel's order --
// wavefront 0 first, wavefront 1 CONTINUES from wavefront 0's per-lane partials (handed over through LDS) and forms the wave-wide
// sum -- so the two kernels leave the same bits, and with them the same schedule decisions (tests/test_gpu_batch.py compares the
// one-launch-per-iteration kernels with the persistent one-wavefront launches bit for bit).
#pragma once
#include <hip/hip_runtime.h>

#include "psd_sign_closed.h"

namespace cuadmm {

// the two wavefronts' tiles (NT = 3): row block / column block of tile t of role w
__host__ __device__ constexpr int sw2_ti(int w, int t) { return w == 0 ? 0 : (t < 2 ? 1 : 2); }
__host__ __device__ constexpr int sw2_tj(int w, int t) { return w == 0 ? t : (t == 0 ? 1 : 2); }

constexpr int kSw2Xch = 3 * 64 + 8;           // doubles of the exchange area behind the tile: three per-lane partials + a few scalars
constexpr size_t kSw2LdsBytes = sizeof(double) * (SignWaveT<3>::NP * SignWaveT<3>::LD + kSw2Xch);

// element (row, col) of the symmetric matrix stored on and above the diagonal (row block b of `row`, column block x of `col`)
template <int LD>
__device__ __forceinline__ double sw2_sym_read(const double* __restrict__ M, int b, int x, int row, int col) {
  if (x > b) return M[row * LD + col];
  if (x < b) return M[col * LD + row];
  return M[swc_sym<LD>(row, col)];
}

template <int ROLE>
__device__ __forceinline__ void psd_sign_closed2_role(const ClosedArgs& fz, int n, double* S, double* xch, int* steps_out, int* hint,
                                                      long long off, int slot, long long poff, int hdr, SwcKArg ka) {
  constexpr int NT = 3;
  using Cfg = SignWaveT<NT>;
  constexpr int LD = Cfg::LD, NP = Cfg::NP;
  constexpr int U = 10;                                   // slots of the flat walk per wavefront: 0 .. 9 | 10 .. 19 (19 in use)
  constexpr int X0 = ROLE == 0 ? 0 : 1;                   // first column block whose fragments this role holds
  const int lane = lane_id();
  const int r16 = lane & 15, kk = lane >> 4;
  const int len = n * (n + 1) / 2;
  const int ksteps = (n + 3) >> 2;
  const bool upd = fz.mode == 0;
  const ClosedRec* __restrict__ rec = fz.rec + slot;
  const int e0 = 64 * U * ROLE + lane;                    // this lane's first svec slot
  const unsigned* __restrict__ tabl = g_swc_tab<NT>.v + e0;
  const double* __restrict__ Xl = fz.X + off + e0;
  const double* __restrict__ Cl = fz.C + off + e0;
  auto at = [&](unsigned byte_off) -> double& { return *reinterpret_cast<double*>(reinterpret_cast<char*>(S) + byte_off); };
  double* xl = xch + lane;                                // per-lane exchange slots: xl[0], xl[64], xl[128]; scalars from xch[192]

  // ---- trip 2 (psd_sign_closed.h): both wavefronts their half of X and C; wavefront 0 the record and the rows' old A X, A (S - C)
  const int nk = closed_hdr_nk(hdr), nnz = closed_hdr_nnz(hdr), nrounds = closed_hdr_nrounds(hdr);
  const int l8 = lane & 7;
  const bool mine = lane < nk;
  unsigned tb[U];
  double xv[U], cv[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { tb[u] = tabl[64 * u]; xv[u] = Xl[64 * u]; cv[u] = Cl[64 * u]; }
  double nzv = 0.0, dk = 1.0, bk = 0.0, ax_old = 0.0, as_old = 0.0;
  unsigned nzt = 0;
  int nzrk = 0, row = 0;
  double lrow[kClosedMaxRows], lcol[kClosedMaxRows];
  if (ROLE == 0) {
    nzv = rec->v[lane]; nzt = rec->nzt[lane]; nzrk = rec->rk[lane]; row = rec->rows[l8];
    dk = rec->D[l8]; bk = rec->b[l8];
    ax_old = fz.cl_out[16 * (long long)slot + l8]; as_old = fz.cl_out[16 * (long long)slot + 8 + l8];
#pragma unroll
    for (int q = 0; q < kClosedMaxRows; ++q) { lrow[q] = rec->L[l8 * kClosedMaxRows + q]; lcol[q] = rec->L[q * kClosedMaxRows + l8]; }
  }
  {                                                        // the tile starts at zero (padding; the Rd1 slots A^T y is summed into)
    sl_v2f64* S2 = reinterpret_cast<sl_v2f64*>(S);
#pragma unroll 1
    for (int i = 64 * ROLE + lane; i < NP * LD / 2; i += 128) S2[i] = sl_v2f64{0.0, 0.0};
  }
  __syncthreads();
  double pby_sum = 0.0;
  if (ROLE == 0) {
    // y_B = (L D L^T)^-1 rhs_B and A^T y scattered into the Rd1 slots: wavefront 0, exactly as in psd_sign_closed.h
    double yk = 0.0;
    if (nk > 0) {
      const double rp = __dadd_rn(-ax_old, bk);
      double x = mine ? __dadd_rn(-as_old, __dmul_rn(fz.isig, rp)) : 0.0;
#pragma unroll
      for (int j = 0; j < kClosedMaxRows - 1; ++j) { const double xj = sw_readlane(x, j); x = __dsub_rn(x, __dmul_rn(lrow[j], xj)); }
      double yv = x / dk;
#pragma unroll
      for (int i = kClosedMaxRows - 1; i >= 1; --i) { const double yi = sw_readlane(yv, i); yv = __dsub_rn(yv, __dmul_rn(lcol[i], yi)); }
      yk = mine ? yv : 0.0;
      if (mine) fz.y_out[row] = yv;
    }
    pby_sum = upd ? swc_uniform(wave_sum(mine ? bk * yk : 0.0)) : 0.0;
    const double yq = __shfl(yk, nzrk & 7, 64);
    const int myround = nzrk >> 3;
    for (int rd = 0; rd < nrounds; ++rd) {
      if (lane < nnz && myround == rd) at(nzt >> 16) = fma(nzv, yq, at(nzt >> 16));
      wave_fence();
    }
  }
  __syncthreads();
  // ---- the flat walk, slots 0 .. 9 | 10 .. 18: Rd1 = A^T y - C (kept), Xb = X + sigma Rd1 -> the upper triangle; || Xb ||_F^2
  double ss = 0.0;
  unsigned okm = 0;                                        // bit u: slot u of this lane is a real element
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e_ = e0 + 64 * u;
    const bool ok_ = e_ < len;
    const unsigned up = tb[u] & 0xfff8u, lo = tb[u] >> 16;
    const double r1 = at(lo) - cv[u];
    const double xb = xv[u] + r1 * fz.sig;
    if (ok_) {
      at(lo) = r1;
      if (!upd) fz.Rd1[off + e_] = r1;
      at(up) = (tb[u] & 1u) ? xb : xb * kSqrt2Inv;
      okm |= 1u << u;
    }
    xv[u] = xb * xb;                                       // the term of the norm, added below in slot order
  }
  if (ROLE == 0) {
#pragma unroll
    for (int u = 0; u < U; ++u) if (okm & (1u << u)) ss += xv[u];
    xl[0] = ss;
  }
  __syncthreads();
  if (ROLE == 1) {                                         // wavefront 1 continues wavefront 0's per-lane sum
    ss = xl[0];
#pragma unroll
    for (int u = 0; u < U; ++u) if (okm & (1u << u)) ss += xv[u];
    const double tot = wave_sum(ss);
    if (lane == 0) xch[192] = tot;
  }
  __syncthreads();
  const double nrm = sqrt(swc_uniform(xch[192]));
  const double scale = nrm > 0.0 ? 1.0 / nrm : (nrm == 0.0 ? 0.0 : nrm);
  SignSched sched;
  if (hint) {
    int h = __builtin_amdgcn_readfirstlane(*hint);
    if (h > 1 && (((unsigned)fz.iter0 + (unsigned)slot) & 15u) == 15u) --h;
    if (h > 0) sched.lift0 = h;
  }
  double f[4 * NT][NT];                                    // this role's fragments: column blocks X0 .. 2
  bool last = false;
  while (!last) {
    // ---- fragments f[s][x] = S[4 s + kk][16 x + r16] of the column blocks this role multiplies with
#pragma unroll
    for (int s = 0; s < 4 * NT; ++s)
#pragma unroll
      for (int x = X0; x < NT; ++x) f[s][x] = sw2_sym_read<LD>(S, s / 4, x, 4 * s + kk, 16 * x + r16);
    if (sched.steps == 0) {
#pragma unroll
      for (int s = 0; s < 4 * NT; ++s)
#pragma unroll
        for (int x = X0; x < NT; ++x) f[s][x] *= scale;
    }
    __syncthreads();                                       // B1: the upper storage becomes Y
    // ---- Y = S S on this role's tiles
    sl_v4f64 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[t] = sl_v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4 * NT; ++s)
      if (s < ksteps) {
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[s][sw2_ti(ROLE, t)], f[s][sw2_tj(ROLE, t)], acc[t], 0, 0, 0);
      }
    const bool stats = sched.needs_stats(), stats_ab = stats && sched.needs_ab();
    double pa = 0.0, pb = 0.0;
    // own tiles of Y -> the upper storage (diagonal tiles: the upper triangle decides)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rowi = 16 * sw2_ti(ROLE, t) + kk + 4 * r, coli = 16 * sw2_tj(ROLE, t) + r16;
        if (sw2_ti(ROLE, t) != sw2_tj(ROLE, t) || coli >= rowi) S[rowi * LD + coli] = acc[t][r];
      }
    // the statistics of Y, per lane, in the one-wavefront kernel's tile order: wavefront 0's part now, handed over at B2
    if (stats_ab && ROLE == 0) {
      pa = 0.0; pb = 0.0;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (sw2_ti(0, t) == sw2_tj(0, t) && kk + 4 * r == r16) pa += acc[t][r];
          pb += (sw2_ti(0, t) == sw2_tj(0, t) ? 1.0 : 2.0) * (acc[t][r] * acc[t][r]);
        }
      xl[0] = pa; xl[64] = pb;
    }
    __syncthreads();                                       // B2: Y complete in LDS
    if (stats_ab && ROLE == 1) {
      pa = xl[0]; pb = xl[64];
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (sw2_ti(1, t) == sw2_tj(1, t) && kk + 4 * r == r16) pa += acc[t][r];
          pb += (sw2_ti(1, t) == sw2_tj(1, t) ? 1.0 : 2.0) * (acc[t][r] * acc[t][r]);
        }
    }
    // ---- Z = S Y on this role's tiles, B operands Y[16 b + 4 s + kk][16 j + r16] from LDS
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[t] = sl_v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (4 * b + s < ksteps) {
          double yb[NT];
#pragma unroll
          for (int j = X0; j < NT; ++j) yb[j] = sw2_sym_read<LD>(S, b, j, 16 * b + 4 * s + kk, 16 * j + r16);
#pragma unroll
          for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[4 * b + s][sw2_ti(ROLE, t)], yb[sw2_tj(ROLE, t)], acc[t], 0, 0, 0);
        }
    double mu;
    if (stats) {
      // || S - S Y ||_F^2 per lane: wavefront 0's tiles, then wavefront 1 continues; wavefront 1 forms the wave-wide sums
      double pg = 0.0;
      if (ROLE == 0) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double d = f[4 * sw2_ti(0, t) + r][sw2_tj(0, t)] - acc[t][r];
            pg += (sw2_ti(0, t) == sw2_tj(0, t) ? 1.0 : 2.0) * (d * d);
          }
        xl[128] = pg;
      }
      __syncthreads();                                     // B5 (also: every read of Y is done)
      if (ROLE == 1) {
        pg = xl[128];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double d = f[4 * sw2_ti(1, t) + r][sw2_tj(1, t)] - acc[t][r];
            pg += (sw2_ti(1, t) == sw2_tj(1, t) ? 1.0 : 2.0) * (d * d);
          }
        double ta = 0.0, tbv = 0.0;
        if (stats_ab) { ta = wave_sum(pa); tbv = wave_sum(pb); }
        const double tg = wave_sum(pg);
        if (lane == 0) { xch[193] = ta; xch[194] = tbv; xch[195] = tg; }
      }
      __syncthreads();                                     // B6
      mu = sched.decide<false>(n, swc_uniform(xch[193]), swc_uniform(xch[194]), swc_uniform(xch[195]), last);
    } else {
      mu = sched.decide<false>(n, 0.0, 0.0, 0.0, last);
      __syncthreads();                                     // B3: every read of Y is done
    }
    const double alpha = -0.5 * mu * mu * mu, beta = 1.5 * mu;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double z = alpha * acc[t][r] + beta * f[4 * sw2_ti(ROLE, t) + r][sw2_tj(ROLE, t)];
        const int rowi = 16 * sw2_ti(ROLE, t) + kk + 4 * r, coli = 16 * sw2_tj(ROLE, t) + r16;
        if (sw2_ti(ROLE, t) != sw2_tj(ROLE, t) || coli >= rowi) S[rowi * LD + coli] = z;
      }
    __syncthreads();                                       // B4: the next iterate is complete
  }
  if (ROLE == 0 && lane == 0) {
    if (steps_out) *steps_out = sched.steps;
    if (hint) *hint = sched.lifts;
  }
  // ---- epilogue: fragments of the sign matrix, Xb rebuilt from X (read again) and the resident Rd1, P = (Xb + S Xb) / 2
  const ClosedArgs fe = swc_args<true>(fz, ka);
  const double* __restrict__ Xl_e = fe.X + off + e0;
  const double* __restrict__ Cl_e = fe.C + off + e0;
#pragma unroll
  for (int u = 0; u < U; ++u) { tb[u] = tabl[64 * u]; xv[u] = Xl_e[64 * u]; }
#pragma unroll
  for (int s = 0; s < 4 * NT; ++s)
#pragma unroll
    for (int x = X0; x < NT; ++x) f[s][x] = sw2_sym_read<LD>(S, s / 4, x, 4 * s + kk, 16 * x + r16);
  __syncthreads();                                         // E1
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e_ = e0 + 64 * u;
    const unsigned up = tb[u] & 0xfff8u, lo = tb[u] >> 16;
    const double xb = xv[u] + at(lo) * fe.sig;
    if (e_ < len) at(up) = (tb[u] & 1u) ? xb : xb * kSqrt2Inv;
  }
  __syncthreads();                                         // E2: Xb complete
  {
    sl_v4f64 p[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) p[t][r] = sw2_sym_read<LD>(S, sw2_ti(ROLE, t), sw2_tj(ROLE, t), 16 * sw2_ti(ROLE, t) + kk + 4 * r, 16 * sw2_tj(ROLE, t) + r16);
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (4 * b + s < ksteps) {
          double yb[NT];
#pragma unroll
          for (int j = X0; j < NT; ++j) yb[j] = sw2_sym_read<LD>(S, b, j, 16 * b + 4 * s + kk, 16 * j + r16);
#pragma unroll
          for (int t = 0; t < 3; ++t) p[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[4 * b + s][sw2_ti(ROLE, t)], yb[sw2_tj(ROLE, t)], p[t], 0, 0, 0);
        }
#pragma unroll
    for (int u = 0; u < U; ++u) { xv[u] = Xl_e[64 * u]; cv[u] = Cl_e[64 * u]; }
    __syncthreads();                                       // E3: every read of Xb is done
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rowi = 16 * sw2_ti(ROLE, t) + kk + 4 * r, coli = 16 * sw2_tj(ROLE, t) + r16;
        if (sw2_ti(ROLE, t) != sw2_tj(ROLE, t) || coli >= rowi) S[rowi * LD + coli] = 0.5 * p[t][r];
      }
  }
  __syncthreads();                                         // E4: P complete
  // ---- the projection leaves through the flat walk: S, Rd, X updates; S - C and the new X staged for the block's rows
  const ClosedArgs fw = swc_args<true>(fz, ka);
  bool bad = false;
  double* __restrict__ Sg = fw.S + off + e0;
  double* __restrict__ Xg = fw.X + off + e0;
  double trd[U], tcx[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int e_ = e0 + 64 * u;
    const bool ok_ = e_ < len;
    const unsigned up = tb[u] & 0xfff8u, lo = tb[u] >> 16;
    const double pm = at(up), r1 = at(lo);
    bad |= ok_ && !(fabs(pm) <= 1.7976931348623157e308);
    const double xp = (tb[u] & 1u) ? pm : pm * kSqrt2;
    const double x = xv[u];
    const double xdiff = xp - x;
    const double sv = fw.inv_sig * xdiff - r1;
    double xn = x, rd = 0.0;
    if (upd) { rd = r1 + sv; xn = x + fw.tau_sig * rd; }
    trd[u] = rd * rd; tcx[u] = cv[u] * xn;
    if (ok_) {
      Sg[64 * u] = sv;
      if (upd) Xg[64 * u] = xn;
      at(up) = sv - cv[u];
      at(lo) = xn;
    }
  }
  double s_rd = 0.0, s_cx = 0.0;
  if (upd && ROLE == 0) {
#pragma unroll
    for (int u = 0; u < U; ++u) if (okm & (1u << u)) { s_rd += trd[u]; s_cx += tcx[u]; }
    xl[0] = s_rd; xl[64] = s_cx;
  }
  if (bad && fw.fail) atomicAdd(fw.fail, 1);
  __syncthreads();                                         // E5: the staging is complete
  if (ROLE == 1) {
    if (upd) {
      s_rd = xl[0]; s_cx = xl[64];
#pragma unroll
      for (int u = 0; u < U; ++u) if (okm & (1u << u)) { s_rd += trd[u]; s_cx += tcx[u]; }
      s_rd = wave_sum(s_rd);
      s_cx = wave_sum(s_cx);
      if (lane == 0) { fw.partials[poff + 2 * (long long)slot] = s_rd; fw.partials[poff + 2 * (long long)slot + 1] = s_cx; }
    }
  } else {
    // the block's constraint rows: one lane per nonzero, one lane per row adds its segment in order (psd_sign_closed.h)
    const ClosedRec* __restrict__ rec_e = fw.rec + slot;
    const double nzv_e = rec_e->v[lane];
    const unsigned nzt_e = rec_e->nzt[lane];
    const int row_e = rec_e->rows[l8];
    const double bk_e = rec_e->b[l8];
    const double nrmA_e = rec_e->normA[l8];
    const int kb = rec_e->nzp[l8], ke = rec_e->nzp[l8 + 1];
    const double ps = nzv_e * at(nzt_e & 0xfff8u);
    const double px = nzv_e * at(nzt_e >> 16);
    const int maxlen = closed_hdr_maxlen(hdr);
    double as = 0.0, ax = 0.0;
    for (int t = 0; t < maxlen; ++t) {
      const int src = (kb + t) & 63;
      const double vs = __shfl(ps, src, 64), vx = __shfl(px, src, 64);
      if (mine && kb + t < ke) { as += vs; ax += vx; }
    }
    if (mine) {
      fw.outS[row_e] = as;
      fw.cl_out[16 * (long long)slot + 8 + lane] = as;
      if (upd) { fw.outX[row_e] = ax; fw.cl_out[16 * (long long)slot + lane] = ax; }
    }
    if (upd) {
      const double ro = nrmA_e * (bk_e - ax) * fw.bscale;
      double pr = mine ? ro * ro : 0.0;
      pr = wave_sum(pr);
      if (lane == 0) { fw.partials2[poff + 2 * (long long)slot] = pr; fw.partials2[poff + 2 * (long long)slot + 1] = pby_sum; }
    }
  }
}

}  // namespace cuadmm
