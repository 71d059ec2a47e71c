// One whole ADMM iteration of a CLOSED block inside the projection kernel, in THREE memory round trips.
//
// A block is closed when every constraint that touches it is one of its own local rows (at most kClosedMaxRows of them, at most
// kFuseRowsMax nonzeros in total): the block-diagonal problems, BASELINE configs[1].  Such a block needs nothing from outside
// except sigma and tau: it solves for its own multipliers, forms A^T y, Rd1, Xb, projects, updates S and X, evaluates its rows of
// A X and A (S - C) and leaves its share of the four scalars of the stopping test (reference src/solver.cu:478-527,534-656,
// 746-776 for one block).
//
// Why this file exists (measured with the tick stamps of CUADMM_CU_DBG at full occupancy, 10 000 x 32 x 32, round 3): in the
// generic fused body (psd_sign_wave.h) a task spent 70 k ticks in its prologue, 108 k in the Newton-Schulz iteration and 48 k
// in its epilogue -- the matrix cores idle for half of a wavefront's lifetime -- because a DEPENDENT memory round trip costs
// ~6 k ticks (2.5 us) on the loaded chip and the body made about ten of them: descriptor -> local-row descriptor -> row indices
// -> right-hand side / factor -> svec + row pointers -> two or three rounds of the CSR gather loop; then X and Rd1 again to
// rebuild Xb, then X, Rd1, C once more for the updates.  Here:
//
//   trip 1  the block descriptor (scalar load);
//   trip 2  everything the prologue needs, addressed from the descriptor alone: the block's RECORD (ClosedRec: its rows, their
//           b / normA / D, the dense factor of its diagonal block of A A^T, its <= 64 nonzeros of A), its rows of
//           [A X | A (S - C)] from a per-block array (cl_out: no row indirection), X and C along the flat svec walk;
//   trip 3  in the epilogue: X and C again (the registers are needed by the iteration in between).
//
// A^T y is SCATTERED through LDS from the block's own nonzeros (one lane per nonzero; nonzeros that hit the same svec slot are
// applied in successive rounds, rows ascending: the summation order of the CSR gather) instead of gathered through row pointers.
// Rd1 = A^T y - C lives in the STRICT LOWER TRIANGLE of the LDS tile (diagonal: the pad column) for the whole task: the
// iterate is symmetric, so it is stored once, on and above the diagonal, and read back with swapped indices below it (one
// store per element instead of two; the transposition scratch of the lower sub-tiles of Y aliases the matching UPPER sub-tile's
// storage).  So Xb is rebuilt from X and the resident Rd1 (same expression, same bits), Rd1 never goes to HBM in mode 0
// (68 -> 52 bytes per svec element), and the final product starts its accumulators at Xb:  P = (Xb + S Xb) / 2.
#pragma once
#include <hip/hip_runtime.h>

#include "psd_sign_wave.h"

namespace cuadmm {

// element (row, col) of a symmetric matrix stored on and above the diagonal
template <int LD>
__device__ __forceinline__ int swc_sym(int row, int col) { return row <= col ? row * LD + col : col * LD + row; }
// where Rd1 of the element (r, c), r <= c, of the upper triangle lives: strictly below the diagonal, diagonal in the pad column
template <int LD, int NP>
__device__ __forceinline__ int swc_low(int r, int c) { return r == c ? r * LD + NP : c * LD + r; }

// operand fragments f[s][x] = M[4 s + kk][16 x + r16] of a symmetric matrix stored on and above the diagonal
template <int NT>
__device__ __forceinline__ void swc_frags(const double* __restrict__ M, int r16, int kk, double (&f)[4 * NT][NT]) {
  constexpr int LD = SignWaveT<NT>::LD;
#pragma unroll
  for (int s = 0; s < 4 * NT; ++s)
#pragma unroll
    for (int x = 0; x < NT; ++x) {
      const int b = s / 4, row = 4 * s + kk, col = 16 * x + r16;
      if (x > b) f[s][x] = M[row * LD + col];
      else if (x < b) f[s][x] = M[col * LD + row];
      else f[s][x] = M[swc_sym<LD>(row, col)];
    }
}
// the same matrix as sub-tiles in accumulator layout: t[b][j][r] = M[16 b + kk + 4 r][16 j + r16]
template <int NT>
__device__ __forceinline__ void swc_tiles(const double* __restrict__ M, int r16, int kk, sl_v4f64 (&t)[NT][NT]) {
  constexpr int LD = SignWaveT<NT>::LD;
#pragma unroll
  for (int b = 0; b < NT; ++b)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * b + kk + 4 * r, col = 16 * j + r16;
        if (j > b) t[b][j][r] = M[row * LD + col];
        else if (j < b) t[b][j][r] = M[col * LD + row];
        else t[b][j][r] = M[swc_sym<LD>(row, col)];
      }
}
// upper sub-tiles -> storage on and above the diagonal
template <int NT>
__device__ __forceinline__ void swc_store_upper(double* __restrict__ M, int r16, int kk, const sl_v4f64 (&d)[NT][NT]) {
  constexpr int LD = SignWaveT<NT>::LD;
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = i; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * i + kk + 4 * r, col = 16 * j + r16;
        if (i != j || col >= row) M[row * LD + col] = d[i][j][r];
      }
}
// lower sub-tiles of a symmetric matrix whose upper sub-tiles sit in accumulator layout: t[j][i] = t[i][j]^T (i < j), transposed
// through the storage of the UPPER sub-tile (i, j) of the tile (dead while the fragments of the iterate are in registers)
template <int NT>
__device__ __forceinline__ void swc_lower_write(double* __restrict__ M, int r16, int kk, const sl_v4f64 (&t)[NT][NT]) {
  constexpr int LD = SignWaveT<NT>::LD;
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = i + 1; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) M[(16 * i + r16) * LD + 16 * j + kk + 4 * r] = t[i][j][r];   // element (kk + 4 r, r16) -> scr[r16][kk + 4 r]
  wave_fence();
}
template <int NT>
__device__ __forceinline__ void swc_lower_read(const double* __restrict__ M, int r16, int kk, sl_v4f64 (&t)[NT][NT]) {
  constexpr int LD = SignWaveT<NT>::LD;
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = i + 1; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) t[j][i][r] = M[(16 * i + kk + 4 * r) * LD + 16 * j + r16];
}

// a wave-uniform double into scalar registers (a VALU result lives in a VGPR even when every lane holds the same bits: across
// the iteration it would cost two of the 128, i.e. a spill)
__device__ __forceinline__ double swc_uniform(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
#else
  return v;
#endif
}
// keeps the slots of a walk apart in the schedule: without it the compiler forms the LDS addresses of all slots up front
// (18 registers) and spills them
__device__ __forceinline__ void swc_sched_split() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0);
#endif
}

// e -> byte offsets into the tile of the element's slot in the upper triangle (bits 0..15; bit 0 set: a diagonal element) and of
// its Rd1 slot (bits 16..31), e = c (c + 1) / 2 + r, r <= c; one table per tile geometry, 64 NSLOT entries (a lane's slots are
// e = lane + 64 u: entries past the last element are never used for an access that matters)
template <int NT>
struct SwcTab {
  static constexpr int N = 64 * SignWaveT<NT>::U * ((SignWaveT<NT>::NSLOT + SignWaveT<NT>::U - 1) / SignWaveT<NT>::U);   // whole batches
  unsigned v[N];
  constexpr SwcTab() : v() {
    constexpr int LD = SignWaveT<NT>::LD, NP = SignWaveT<NT>::NP;
    int e = 0;
    for (int c = 0; c < NP && e < N; ++c)
      for (int r = 0; r <= c && e < N; ++r) {
        const unsigned up = (unsigned)(r * LD + c) * 8u | (r == c ? 1u : 0u);
        const unsigned lo = (unsigned)(r == c ? r * LD + NP : c * LD + r) * 8u;
        v[e++] = up | (lo << 16);
      }
  }
};
template <int NT> __device__ const SwcTab<NT> g_swc_tab = SwcTab<NT>();

// the kernel arguments of the closed-block kernels (kept small: at 100 SGPRs of pointers the compiler spills them into VGPR
// lanes, and every v_writelane / v_readlane is a VALU slot the fp64 matrix pipe pays for)
struct ClosedArgs {
  const PsdDesc* desc;
  int* steps; int* hint; int* fail;
  long long* dbg;
  double* X; double* S; double* Rd1; const double* C;
  const ClosedRec* rec;
  double* cl_out; double* y_out; double* outS; double* outX;
  double* partials; double* partials2;
  double sig, inv_sig, tau_sig, isig, bscale;
  long long pstride;
  int mode, iters, first, count;
  int iter0;                                          // iterations run before this launch (the schedule hints age with it)
  int dcache;                                         // persistent launches: the workgroup keeps its members' descriptors in LDS
};

// The kernel arguments again, from the kernarg segment (persistent launches: scalar loads, cached): a phase of a task takes a
// fresh copy instead of keeping the previous phase's scalar registers alive across the iteration -- at this kernel's budget
// those are spilled into VGPR lanes, and every v_writelane / v_readlane takes ~4.5 cycles from the port the MFMAs issue on.
using SwcKArg = __attribute__((address_space(4))) const char*;
template <bool RELOAD>
__device__ __forceinline__ ClosedArgs swc_args(const ClosedArgs& fz, SwcKArg& ka) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (RELOAD) {
    asm volatile("" : "+s"(ka));
    return *reinterpret_cast<__attribute__((address_space(4))) const ClosedArgs*>(ka);
  }
#endif
  return fz;
}

#define CUADMM_SWC_STAMP(k) \
  if (dbg) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (lane == 0) dbg[k] = (long long)__builtin_readcyclecounter() - c0; }

// off = svec offset of the block, slot = its partial-sum / record slot, poff = offset of this iteration's partial arrays.
// The svec arrays are read up to 64 NSLOT elements past the block's first element WITHOUT a bounds clamp (the engine pads the
// allocations): every slot of a batch then shares one address register and immediate offsets.
// TASK_LOOP: the body is inlined into the task loop of psd_sign_closed_cu_kernel, where everything derived from the lane id is
// loop-invariant -- hoisted out of the loop it would stay live across the whole body; the lane id is made opaque instead.
// FULL: every block of the launch has n = NP (BASELINE configs[1]: 32): the length of the svec range and with it the validity of
// every slot of the walks is known at compile time -- no masks, no branches around the slots.  (On this chip everything a
// wavefront issues on the vector ALU takes its cycles from the port the fp64 MFMAs issue on, tools/ubench/mfma_coissue.hip: an
// integer VALU instruction costs ~2 cycles of matrix-core time, an fp64 one ~4, a v_readlane ~4.5 -- the prologue and epilogue
// of a task are paid in matrix-core cycles, not hidden behind them.)
// DBG: the tick stamps of the developer aid (psd_debug = 2) are compiled into their own instantiation -- a run-time "if (dbg)" keeps
// the stamps' registers and v_writelanes in the production kernel.
template <int NT, bool TASK_LOOP = false, bool FULL = false, bool DBG = false>
__device__ __forceinline__ void psd_sign_closed_body(const ClosedArgs& fz, int n_arg, double* S, int* steps_out, int* hint, long long* dbg_arg,
                                                     long long off, int slot, long long poff, int hdr, int it_local, SwcKArg ka) {
  long long* const dbg = DBG ? dbg_arg : nullptr;
  using Cfg = SignWaveT<NT>;
  constexpr int LD = Cfg::LD, NP = Cfg::NP, U = Cfg::U, NSLOT = Cfg::NSLOT;
  constexpr int NB = (NSLOT + U - 1) / U;                 // batches of the flat walk
  const int n = FULL ? NP : n_arg;
  int lane_p = lane_id();
#if defined(__HIP_DEVICE_COMPILE__)
  if (TASK_LOOP) asm volatile("" : "+v"(lane_p));
  __builtin_assume(lane_p >= 0 && lane_p < 64);
#endif
  const int lane = lane_p;
  const int r16 = lane & 15, kk = lane >> 4;
  const int len = n * (n + 1) / 2;
  const int ksteps = FULL ? 4 * NT : ((n + 3) >> 2);      // k-steps with at least one real row
  const long long c0 = dbg ? (long long)__builtin_readcyclecounter() : 0;
  const bool upd = fz.mode == 0;
  const ClosedRec* __restrict__ rec = fz.rec + slot;
  const unsigned* __restrict__ tabl = g_swc_tab<NT>.v + lane;
  const double* __restrict__ Xl = fz.X + off + lane;
  const double* __restrict__ Cl = fz.C + off + lane;
  auto at = [&](unsigned byte_off) -> double& { return *reinterpret_cast<double*>(reinterpret_cast<char*>(S) + byte_off); };

  // ---- trip 2: the record, the block's rows of [A X | A (S - C)], and the first batch of table / X / C.  Every address is known
  // from the descriptor (the record's header rides in PsdDesc::pad[0]: closed_hdr_*), and no load is conditional on a loaded
  // value -- the first version read nk from the record and only then issued the loads of the factor, X and C: two dependent
  // round trips of ~6 k ticks each where one does
  const int nk = closed_hdr_nk(hdr), nnz = closed_hdr_nnz(hdr), nrounds = closed_hdr_nrounds(hdr);   // wave-uniform
  const int l8 = lane & 7;
  const bool mine = lane < nk;
  const double nzv = rec->v[lane];
  const unsigned nzt = rec->nzt[lane];
  const int nzrk = rec->rk[lane];
  const int row = rec->rows[l8];
  const double dk = rec->D[l8], bk = rec->b[l8];
  const double ax_old = fz.cl_out[16 * (long long)slot + l8], as_old = fz.cl_out[16 * (long long)slot + 8 + l8];
  double lrow[kClosedMaxRows], lcol[kClosedMaxRows];
#pragma unroll
  for (int q = 0; q < kClosedMaxRows; ++q) {      // rows / columns l8 of the factor for every lane (lanes >= 8 repeat them): masked below
    lrow[q] = rec->L[l8 * kClosedMaxRows + q];                                            // L[lane][q]
    lcol[q] = rec->L[q * kClosedMaxRows + l8];                                            // L[q][lane]
  }
  unsigned tb[U];
  double xv[U], cv[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { tb[u] = tabl[64 * u]; xv[u] = Xl[64 * u]; cv[u] = Cl[64 * u]; }
  // (no masks on the factor: the record holds the STRICT lower triangle and zeros everywhere else, rows and columns >= nk
  // included, and D = 1 there -- lanes >= 8 repeat lanes 0..7 and are never read)
  // the whole tile starts at zero: the padding of a block smaller than the tile, and the Rd1 slots A^T y is summed into
  {
    sl_v2f64* S2 = reinterpret_cast<sl_v2f64*>(S);
    double zero = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
    if (TASK_LOOP) asm volatile("" : "+v"(zero));            // formed here: hoisted out of the task loop the four registers are spilled
#endif
#pragma unroll 1
    for (int i = lane; i < NP * LD / 2; i += 64) S2[i] = sl_v2f64{zero, zero};
    wave_fence();
  }
  // y_B = (L D L^T)^-1 rhs_B, one lane per row, the serial order and the unfused arithmetic of forest_solve_kernel
  double yk = 0.0;
  if (nk > 0) {
    const double rp = __dadd_rn(-ax_old, bk);                                             // Rp = -A X + b
    double x = mine ? __dadd_rn(-as_old, __dmul_rn(fz.isig, rp)) : 0.0;
    // both sweeps column by column: 7 + 7 broadcasts from a compile-time lane (v_readlane; the first version's row-oriented
    // backward sweep needed 28 dependent ds_bpermute round trips), unfused multiply and subtract as in forest_solve_kernel
#pragma unroll
    for (int j = 0; j < kClosedMaxRows - 1; ++j) {                                        // L z = rhs
      const double xj = sw_readlane(x, j);
      x = __dsub_rn(x, __dmul_rn(lrow[j], xj));                                           // lrow[j] = 0 for lanes <= j
    }
    double yv = x / dk;                                                                   // D^-1, then L^T y = z
#pragma unroll
    for (int i = kClosedMaxRows - 1; i >= 1; --i) {
      const double yi = sw_readlane(yv, i);                                               // final: rows > i are done
      yv = __dsub_rn(yv, __dmul_rn(lcol[i], yi));                                         // lcol[i] = L[i][lane] = 0 for lanes >= i
    }
    yk = mine ? yv : 0.0;
    if (mine) fz.y_out[row] = yv;
  }
  // the rows' share of b^T y (rp_stats_partial_kernel's expression) while y is at hand: wave-uniform, so it waits in scalar
  // registers for the epilogue (the first version re-read the row indices and then y after the iteration: two dependent loads)
  const double pby_sum = upd ? swc_uniform(wave_sum(mine ? bk * yk : 0.0)) : 0.0;
  CUADMM_SWC_STAMP(4);
  // ---- A^T y scattered into the Rd1 slots, one round per multiplicity of an svec slot (rows ascending: the CSR gather's order)
  {
    const double yq = __shfl(yk, nzrk & 7, 64);
    const int myround = nzrk >> 3;
    for (int rd = 0; rd < nrounds; ++rd) {
      if (lane < nnz && myround == rd) at(nzt >> 16) = fma(nzv, yq, at(nzt >> 16));
      wave_fence();
    }
  }
  // ---- the flat walk: Rd1 = A^T y - C (kept in the tile), Xb = X + sigma Rd1 -> the upper triangle; ||Xb||_F on the way
  double ss = 0.0;
#pragma unroll 1
  for (int bt = 0; bt < NB; ++bt) {
    const int base = 64 * U * bt;
    if (bt > 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) { tb[u] = tabl[base + 64 * u]; xv[u] = Xl[base + 64 * u]; cv[u] = Cl[base + 64 * u]; }
    }
    if (bt == 0) { CUADMM_SWC_STAMP(5); }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e_ = base + 64 * u + lane;
      const bool ok_ = (FULL && base + 64 * u + 63 < Cfg::MAXLEN) || e_ < len;
      const unsigned up = tb[u] & 0xfff8u, lo = tb[u] >> 16;
      const double r1 = at(lo) - cv[u];
      const double xb = xv[u] + r1 * fz.sig;
      if (ok_) {
        at(lo) = r1;
        if (!upd) fz.Rd1[off + e_] = r1;                 // mode 1: the stand-alone kernels of the step read it
        ss += xb * xb;
        at(up) = (tb[u] & 1u) ? xb : xb * kSqrt2Inv;
      }
    }
  }
  CUADMM_SWC_STAMP(6);
  const double nrm = sqrt(wave_sum(ss));
  const double scale = nrm > 0.0 ? 1.0 / nrm : (nrm == 0.0 ? 0.0 : nrm);   // NaN propagates (flagged at the store)
  wave_fence();
  double f[4 * NT][NT];
  SignSchedPlain sched;                          // closed blocks = block-diagonal synthetic problems: no gap in their spectra, no mega-lift (sign_sched.h)
  if (hint) {
    int h = __builtin_amdgcn_readfirstlane(*hint);
    // the hint ages by one lift every 16th iteration, staggered over the blocks: a function of the iteration index alone, so one
    // launch per iteration and several iterations per launch run the same schedules
    if (h > 1 && (((unsigned)fz.iter0 + (unsigned)it_local + (unsigned)slot) & 15u) == 15u) --h;
    if (h > 0) sched.lift0 = h;
  }
  bool last = false;
  const long long c1 = dbg ? (long long)__builtin_readcyclecounter() : 0;
  while (!last) {
    swc_frags<NT>(S, r16, kk, f);
    if (sched.steps == 0) {                                // S_0 = Xb / ||Xb||_F: the scale goes onto the fragments
#pragma unroll
      for (int s = 0; s < 4 * NT; ++s)
#pragma unroll
        for (int x = 0; x < NT; ++x) f[s][x] *= scale;
    }
    wave_fence();                                          // the upper storage is scratch from here to the store of the next iterate
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
    // statistics only where the schedule reads them: nothing on the steps whose scale is fixed in advance, ||S - S Y||^2 alone
    // in the plain / finishing phases, tr Y and ||Y||_F^2 as well on the first step and after a probe (wave-uniform)
    const bool stats = sched.needs_stats(), stats_ab = stats && sched.needs_ab();
    double pa = 0.0, pb = 0.0;
    if (stats_ab) {
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
    swc_lower_write<NT>(S, r16, kk, y);
    swt_mma_regB<NT, 0, 1>(f, y, z, ksteps);                        // row block 0 of Y: upper sub-tiles only
    swc_lower_read<NT>(S, r16, kk, y);
    swt_mma_regB<NT, 1, NT>(f, y, z, ksteps);
    double mu;
    if (stats) {
      double ta = 0.0, tbv = 0.0;
      if (stats_ab) { ta = wave_sum(pa); tbv = wave_sum(pb); }
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
      mu = sched.decide<false>(n, ta, tbv, tg, last);
    } else {
      mu = sched.decide<false>(n, 0.0, 0.0, 0.0, last);
    }
    double alpha, beta;
    sched.coefs(mu, alpha, beta);
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = i; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) z[i][j][r] = fma(alpha, z[i][j][r], beta * f[4 * i + r][j]);   // one contraction, spelled out: every instantiation rounds alike
    wave_fence();                                            // the transposition scratch has been read
    swc_store_upper<NT>(S, r16, kk, z);
    wave_fence();
  }
  if (steps_out && lane == 0) *steps_out = sched.steps;
  if (hint && lane == 0) *hint = sched.lifts;
  const long long c2 = dbg ? (long long)__builtin_readcyclecounter() : 0;
  // ---- epilogue.  Trip 3: X (and, behind the final product, C and the block's nonzeros) again; Xb is rebuilt from X and the
  // resident Rd1 with the prologue's expression (same bits) into the upper triangle, once the fragments of S are in registers.
  // Nothing lane-dependent of the prologue is kept across the iteration (it would be spilled at 128 registers): the lane id is
  // made opaque here, so that the pointers are formed again, and the record's fields are read again with the X loads.
  int lane_e = lane;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(lane_e));
#endif
  const unsigned* __restrict__ tabl_e = g_swc_tab<NT>.v + lane_e;
  const ClosedArgs fe = swc_args<true>(fz, ka);      // a fresh copy: none of the prologue's scalar registers lives across the iteration
  const ClosedRec* __restrict__ rec_e = fe.rec + slot;
  const double* __restrict__ Xl_e = fe.X + off + lane_e;
  const double* __restrict__ Cl_e = fe.C + off + lane_e;
  const int l8e = lane_e & 7;
  const bool mine_e = lane_e < nk;
#pragma unroll
  for (int u = 0; u < U; ++u) { tb[u] = tabl_e[64 * u]; xv[u] = Xl_e[64 * u]; }
  swc_frags<NT>(S, r16, kk, f);
  wave_fence();
#pragma unroll 1
  for (int bt = 0; bt < NB; ++bt) {
    const int base = 64 * U * bt;
    if (bt > 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) { tb[u] = tabl_e[base + 64 * u]; xv[u] = Xl_e[base + 64 * u]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e_ = base + 64 * u + lane_e;
      const unsigned up = tb[u] & 0xfff8u, lo = tb[u] >> 16;
      const double xb = xv[u] + at(lo) * fe.sig;
      if ((FULL && base + 64 * u + 63 < Cfg::MAXLEN) || e_ < len) at(up) = (tb[u] & 1u) ? xb : xb * kSqrt2Inv;     // the zero padding of the prologue is still in place
    }
  }
  wave_fence();
  CUADMM_SWC_STAMP(7);
  if (NB > 1) {                                              // several batches: the first one's table again for the final walk
#pragma unroll
    for (int u = 0; u < U; ++u) tb[u] = tabl_e[64 * u];
  } else {
    // one batch: the table entries stay, but the LDS addresses formed from them are NOT kept across the final product for the
    // final walk (18 registers: spilled at this kernel's budget) -- the entries are made opaque, two VALU operations re-form an address
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int u = 0; u < U; ++u) asm volatile("" : "+v"(tb[u]));
#endif
  }
  {
    sl_v4f64 p[NT][NT];
    swc_tiles<NT>(S, r16, kk, p);                            // Xb in accumulator layout: B operand AND initial accumulator
    {
      sl_v4f64 xb[NT][NT];
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int j = 0; j < NT; ++j) xb[b][j] = p[b][j];
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = i; j < NT; ++j)
              if (4 * b + s < ksteps) p[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[4 * b + s][i], xb[b][j][s], p[i][j], 0, 0, 0);
    }
    // the loads of the final walk's first batch ride behind the matrix pipe (X once more: keeping it live across the product
    // costs spills at 128 registers)
#pragma unroll
    for (int u = 0; u < U; ++u) { xv[u] = Xl_e[64 * u]; cv[u] = Cl_e[64 * u]; }
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = i; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) p[i][j][r] *= 0.5;         // P = (Xb + S Xb) / 2
    wave_fence();
    swc_store_upper<NT>(S, r16, kk, p);
  }
  // what the block's constraint rows need from the record, loaded behind the final product too (nothing here depends on a
  // loaded value; held across the product these registers would be spilled)
  const double nzv_e = rec_e->v[lane_e];
  const unsigned nzt_e = rec_e->nzt[lane_e];
  const int row_e = rec_e->rows[l8e];
  const double bk_e = rec_e->b[l8e];
  const double nrmA_e = rec_e->normA[l8e];
  const int kb = rec_e->nzp[l8e], ke = rec_e->nzp[l8e + 1];
  wave_fence();
  CUADMM_SWC_STAMP(8);
  // ---- the projection leaves through the flat walk: S, Rd, X updates and the two sums (the expressions of post_kernel); the
  // slot of P(r, c) then takes S - C and the slot of Rd1(r, c) the new X -- the staging the block's constraint rows read
  bool bad = false;
  double s_rd = 0.0, s_cx = 0.0;
  const ClosedArgs fw = swc_args<true>(fz, ka);
  double* __restrict__ Sg = fw.S + off + lane_e;
  double* __restrict__ Xg = fw.X + off + lane_e;
#pragma unroll 1
  for (int bt = 0; bt < NB; ++bt) {
    const int base = 64 * U * bt;
    if (bt > 0) {
#pragma unroll
      for (int u = 0; u < U; ++u) { tb[u] = tabl_e[base + 64 * u]; xv[u] = Xl_e[base + 64 * u]; cv[u] = Cl_e[base + 64 * u]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e_ = base + 64 * u + lane_e;
      const bool ok_ = (FULL && base + 64 * u + 63 < Cfg::MAXLEN) || e_ < len;
      const unsigned up = tb[u] & 0xfff8u, lo = tb[u] >> 16;
      const double pm = at(up), r1 = at(lo);
      bad |= ok_ && !(fabs(pm) <= 1.7976931348623157e308);
      const double xp = (tb[u] & 1u) ? pm : pm * kSqrt2;     // Xproj[e]
      const double x = xv[u];
      const double xdiff = xp - x;
      const double sv = fw.inv_sig * xdiff - r1;
      double xn = x;
      double rd = 0.0;
      if (upd) { rd = r1 + sv; xn = x + fw.tau_sig * rd; }
      if (ok_) {
        Sg[base + 64 * u] = sv;
        if (upd) {
          Xg[base + 64 * u] = xn;
          s_rd += rd * rd;
          s_cx += cv[u] * xn;
        }
        at(up) = sv - cv[u];
        at(lo) = xn;
      }
      if ((u % 3) == 2) swc_sched_split();
    }
  }
  CUADMM_SWC_STAMP(9);
  // ---- the block's constraint rows: one lane per nonzero forms a * v from the staging, one lane per row adds its segment in
  // order; with the new A X the rows' share of || Rp ||^2 and b^T y (rp_stats_partial_kernel's expressions)
  {
    wave_fence();
    const double ps = nzv_e * at(nzt_e & 0xfff8u);
    const double px = nzv_e * at(nzt_e >> 16);
    const int maxlen = closed_hdr_maxlen(hdr);                 // longest row of the block
    double as = 0.0, ax = 0.0;
    for (int t = 0; t < maxlen; ++t) {
      const int src = (kb + t) & 63;
      const double vs = __shfl(ps, src, 64), vx = __shfl(px, src, 64);
      if (mine_e && kb + t < ke) { as += vs; ax += vx; }
    }
    if (mine_e) {
      fw.outS[row_e] = as;
      fw.cl_out[16 * (long long)slot + 8 + lane_e] = as;
      if (upd) { fw.outX[row_e] = ax; fw.cl_out[16 * (long long)slot + lane_e] = ax; }
    }
    if (upd) {
      const double ro = nrmA_e * (bk_e - ax) * fw.bscale;
      double pr = mine_e ? ro * ro : 0.0;
      pr = wave_sum(pr);
      const double pby = pby_sum;
      s_rd = wave_sum(s_rd);
      s_cx = wave_sum(s_cx);
      if (lane_e == 0) {
        fw.partials2[poff + 2 * (long long)slot] = pr; fw.partials2[poff + 2 * (long long)slot + 1] = pby;
        fw.partials[poff + 2 * (long long)slot] = s_rd; fw.partials[poff + 2 * (long long)slot + 1] = s_cx;
      }
    }
  }
  if (bad && fw.fail) atomicAdd(fw.fail, 1);
  if (dbg && lane == 0) {   // developer aid (CUADMM_CU_DBG): ticks of prologue / iteration / epilogue, steps
    const long long c3 = (long long)__builtin_readcyclecounter();
    dbg[0] = c1 - c0; dbg[1] = c2 - c1; dbg[2] = c3 - c2; dbg[3] = sched.steps;
  }
}
#undef CUADMM_SWC_STAMP

}  // namespace cuadmm
