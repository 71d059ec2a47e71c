// PSD-cone projection of blocks with n > 64 on the fp64 matrix cores, WITHOUT an eigendecomposition.
//
// The reference sends large blocks one by one to cusolverDnXsyevd and small ones to the batched Jacobi solver, then
// forms V max(L,0) V^T with DGEMMs (src/solver.cu:534-647).  Here, for every block that does not fit the
// register-resident eigensolver (psd_small_reg.h, n <= 64):
//
//     P_+(X) = (X + |X|) / 2,   |X| = X * sign(X),
//
// with the matrix sign function from the Newton-Schulz iteration  S <- p(mu S),  p(x) = 1.5 x - 0.5 x^3,  started
// from S_0 = X / ||X||_1 (all eigenvalues in [-1, 1]; zero eigenvalues stay zero).  Everything is GEMM on
// v_mfma_f64_16x16x4_f64: that is what CDNA4 is good at, while the sequential rotation chains of a tridiagonal
// eigensolver (5 million dependent rotations at n = 2000) are what it is bad at.
//
// Schedule: per block, adaptive, decided ON THE DEVICE (sign_sched.h, lagged variant) -- still no host synchronisation, the
// projection stays stream-ordered.  The products leave tr Y, ||Y||_F^2 and ||S - S Y||_F^2 as per-tile partial sums in
// fixed slots (plain stores: no atomics, no fences inside a launch); the NEXT launch sums the slots in slot order -- every
// workgroup of the S Y product redundantly when there are few tiles, a one-workgroup kernel between the two products when
// there are many -- and runs the state machine, so the sums and with them every decision are bit-reproducible.  The state
// is double-buffered by step parity (readers and the one writer of a launch never touch the same copy).  The launches of a
// finished block return at once.
//   * lift steps (mu = 1.53): eigenvalues in (0, 1] stay in (0, 1] and never fall below p(1.53) = 0.5 once they are
//     there (the gap between the + and - invariant subspaces stays wide: stable), small ones grow by 2.295 per step;
//   * probe / plain steps: quadratic convergence from [0.5, 1] to 1 within roundoff; the statistics tell whether
//     anything is left unresolved and how far below the basin it is.
//   An eigenvalue below the resolution 1e-13 ||X||_1 contributes an error <= |lambda| to the projection.
//
// Every iterate is a polynomial in X, hence symmetric, and all products are of commuting symmetric matrices.  The
// GEMM kernel uses that twice: the left operand is read transposed (row tile of A = rows k of A, coalesced, no LDS
// transpose), and only the tiles on or above the diagonal are computed and MIRRORED on store.  Mirroring halves the
// flops and keeps the iterate EXACTLY symmetric -- with an independently computed lower triangle the skew-symmetric
// rounding error doubles every step (measured: divergence after ~60 steps).
//
// Batching: blocks are grouped by padded size N (multiple of 64); one launch covers a whole group (blockIdx.y =
// member), so many mid-size blocks (n = 65..200) fill the chip the same way one n = 2000 block does.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "device_util.h"
#include "psd_device.h"
#include "psd_large.h"
#include "sign_sched.h"
#include "wave_reduce.h"

namespace cuadmm {

typedef double lg_v4f64 __attribute__((ext_vector_type(4)));
constexpr int LG_TM = 64;   // measured on MI355X: 64x64 tiles (4 workgroups per CU) beat 128x128 at every N (48 vs 41 TFLOP/s at 2048)

// C = alpha * A*B + beta * E over a batch (blockIdx.y = matrix, stride N*N).  A (and, when MIRROR, the product) symmetric.
// MIRROR: blockIdx.x enumerates the tiles (by <= bx) of the upper triangle; C[row][col] and C[col][row] are both written.
// TM = 64: 4 MFMA tiles per wavefront (16 flop per byte of operand traffic); TM = 32: one MFMA tile per wavefront, four
// times as many workgroups -- for matrices whose 64 x 64 tiles would leave most of the 256 CUs idle (N ~ 1000: 136 tiles).
// Per-member device state of the adaptive schedule, double-buffered by step parity: step k reads version k at
// st[(k & 1) * count + member] and its single writer stores version k + 1 into the other copy.
struct SignDevState {
  SignSched sched;
  double mu;                // of the step that produced this version
  int n;                    // true block size
};
// written once per projection (by the writer of the last step): a launch of step j skips the member iff done_at <= j, which
// reads the same whether a concurrent workgroup of step done_at - 1 sees the old or the new value
struct SignDone { int done_at; int steps; };
struct SignArgs {
  SignDevState* st;         // 2 * count
  SignDone* done;           // count
  double* p1;               // [member][tile][2]: tr Y, ||Y||_F^2 of the current step
  double* p2;               // [parity][member][tile]: ||S - S Y||_F^2 of the step with that parity
  int* group;               // group[0]: members not finished, group[1]: largest step count (polled by the host)
  int* hint;                // per block (all blocks of the plan), in/out: schedule warm start; may be null
  const int* ids;           // member -> block id
  int count, step;
  unsigned* bar;            // per member: barrier counter of the one-launch variant (lg_sign_cluster_kernel); may be null
  int* cont;                // [parity][member]: version k says whether step k is the SECOND slot of a clean mega-lift (sign_sched.h); null: never
  int clean;                // this group's schedule takes clean mega-lifts (padded size <= 512: the fifth matrix per member exists)
  int cap;                  // the step limit of this projection (a mega-lift's two slots never straddle it)
};
// fresh schedule state of a member (version 0)
__device__ __forceinline__ void lg_fresh_state(const SignArgs& sa, int m, int id, int n, bool reset_bar = true) {
  SignDevState st;
  st.sched = SignSched();
  if (sa.hint && sa.hint[id] > 0) st.sched.lift0 = sa.hint[id];
  st.sched.clean = sa.clean != 0 && sa.cont != nullptr;
  if (sa.cap > 0 && sa.cap < SignSched::kCap) st.sched.cap = sa.cap;
  st.mu = 1.0;
  st.n = n;
  sa.st[m] = st;
  sa.done[m].done_at = 0x7fffffff;
  sa.done[m].steps = 0;
  if (sa.bar && reset_bar) sa.bar[m] = 0u;
  if (sa.cont) { sa.cont[m] = 0; sa.cont[sa.count + m] = 0; }
}

// Sums the statistics' slots in a fixed order (all 256 threads), runs the schedule's decision for step sa.step on thread 0
// and returns mu to every thread; `writer` stores version step + 1 of the state (exactly one workgroup per member does).
__device__ __forceinline__ double lg_reduce_decide(const SignArgs& sa, int member, int ntiles, bool writer, double* red) {
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int par = sa.step & 1;
  const double* p1 = sa.p1 + (size_t)member * 2 * (size_t)ntiles;
  const double* p2 = sa.p2 + ((size_t)(par ^ 1) * sa.count + member) * (size_t)ntiles;
  // the state travels first, word by word into LDS (thread t: word t): its trip hides behind the slots' -- it used to follow the two barriers --
  // and no thread holds the struct in registers across them
  static_assert(sizeof(SignDevState) % 8 == 0 && sizeof(SignDevState) <= 256, "SignDevState: whole doubles, at most 32");
  constexpr int kStWords = (int)(sizeof(SignDevState) / 8);
  __shared__ double st_sh[32];
  if (tid < kStWords) st_sh[tid] = reinterpret_cast<const double*>(sa.st + ((size_t)par * sa.count + member))[tid];
  double q0 = 0.0, q1 = 0.0, q2 = 0.0;
  for (int t = tid; t < ntiles; t += 256) {
    q0 += p1[2 * t];
    q1 += p1[2 * t + 1];
    if (sa.step > 0) q2 += p2[t];
  }
  q0 = wave_sum(q0);
  q1 = wave_sum(q1);
  q2 = wave_sum(q2);
  __syncthreads();
  if (lane == 0) { red[wave] = q0; red[4 + wave] = q1; red[8 + wave] = q2; }
  __syncthreads();
  if (tid == 0) {
    const double a = (red[0] + red[1]) + (red[2] + red[3]);
    const double b = (red[4] + red[5]) + (red[6] + red[7]);
    const double g2 = (red[8] + red[9]) + (red[10] + red[11]);
    SignDevState v;
    __builtin_memcpy(&v, st_sh, sizeof(SignDevState));
    v.sched.gprev = sa.step == 0 ? -1.0 : sqrt(g2 > 0.0 ? g2 : 0.0);
    bool last;
    v.mu = v.sched.decide<true>(v.n, a, b, 0.0, last);
    red[12] = v.mu;
    red[13] = v.sched.cm;                 // > 0: a mega-lift (sign_sched.h), coefficients -cm, 1 + cm
    red[14] = (double)v.sched.half;       // clean mega-lift: 1 = this step's output is R = S - S Y, 2 = this step applies S + cmc (M - Y M)
    red[15] = v.sched.cmc;
    if (writer) {
      sa.st[(size_t)(par ^ 1) * sa.count + member] = v;
      if (sa.cont) __hip_atomic_store(sa.cont + (size_t)(par ^ 1) * sa.count + member, v.sched.cont ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (last) {
        sa.done[member].steps = v.sched.steps;
        sa.done[member].done_at = sa.step + 1;
        if (sa.hint) sa.hint[sa.ids[member]] = v.sched.lifts;
        atomicMax(sa.group + 1, v.sched.steps);
        atomicSub(sa.group, 1);
      }
    }
  }
  __syncthreads();
  return red[12];
}

// one workgroup per member: the decision of step sa.step as a launch of its own (between Y = S S and S Y) for matrices with
// so many tiles that summing the slots in every workgroup of the S Y product would cost more than a launch
__global__ __launch_bounds__(256) void lg_decide_kernel(SignArgs sa, int ntiles) {
  __shared__ double red[16];
  const int member = (int)blockIdx.x;
  if (sa.done[member].done_at <= sa.step) return;
  (void)lg_reduce_decide(sa, member, ntiles, true, red);
}

// ROLE 0: plain product.  Roles of the sign iteration (MIRROR only; member = blockIdx.y):
// ROLE 1: C = A*A (Y = S S), leaves the slots tr Y, ||Y||_F^2;
// ROLE 2: C = 1.5 mu E - 0.5 mu^3 A*B (T from S, Y), mu decided here by every workgroup (lg_reduce_decide), leaves the
//         slots ||S - S Y||_F^2;   ROLE 4: the same with mu read from the state (lg_decide_kernel ran in between);
// ROLE 5: the final product of the one-launch kernel with the pack kernel's work in its epilogue: Cb = the block's svec in the output vector
//         (not offset by the member), entry (row <= col < n_true) goes to col (col + 1) / 2 + row, off-diagonal ones times sqrt 2 -- no dense copy,
//         no mirror image, no lg_pack_kernel behind it; a non-finite entry raises *fail like that kernel does.
// ROLE 3: the final product: B = the buffer that holds the last iterate (Bb after an even number of steps, B2b after
//         an odd one).
// CLEAN MEGA-LIFT (sign_sched.h; B2b = the member's fifth matrix M in roles 1, 2, 4).  Its first slot is an ordinary step whose coefficients
// (-1, 1) leave R = S - S Y where the next iterate would be; the launches of the SECOND slot find that out from the device state and change
// operands: ROLE 1 forms M = R - R Y instead of Y = S S -- as a FULL product, the tile and its transpose one after the other, no mirroring
// (R (I - Y) is symmetric only up to the noise the step removes) -- and ROLE 2 / 4 forms S + cmc (M - Y M), mirrored, in place over the old
// iterate (which the step before left in the output buffer; an element is read and written by the same thread).
template <bool MIRROR, int TM, int BK>
struct LgGemmCfg {
  static constexpr int LDS = TM + 16;      // row stride (doubles): the 4 k-rows of a fragment read fall on disjoint banks
  static constexpr int SMEM = MIRROR ? (TM * (TM + 1) > 2 * BK * LDS ? TM * (TM + 1) : 2 * BK * LDS) : 2 * BK * LDS;
};
// The body of one output tile: workgroup (tile_x of grid_x, member).  smem: LgGemmCfg::SMEM doubles, red: 16 doubles.
// CLEAN: the instantiation knows the second slot of a clean mega-lift (groups that never take one -- C3's single large block -- run the other: not a register more)
template <bool MIRROR, int TM, int BK, int ROLE, bool CLEAN = false, bool LATE = true>
__device__ __forceinline__ void lg_gemm_sym_body(int N, const double* __restrict__ Ab, const double* __restrict__ Bb,
                                                 double alpha, double beta, const double* __restrict__ Eb,
                                                 double* __restrict__ Cb, int sb, const SignArgs& sg, const double* __restrict__ B2b,
                                                 const int member, const int tile_x, const int grid_x, double* smem, double* red,
                                                 const bool known_live = false, const int n_true = 0, int* fail = nullptr) {
  constexpr int LDS = TM + 16;      // row stride (doubles): the 4 k-rows of a fragment read fall on disjoint banks
  constexpr int WT = TM / 2;        // rows / cols per wave
  constexpr int NTW = WT / 16;      // 16x16 MFMA tiles per wave per direction
  bool slot2 = false;               // ROLE 1: this step is the second slot of a clean mega-lift
  if (ROLE == 1 || ROLE == 2 || ROLE == 4) {
    int sl = 0;
    if (CLEAN && ROLE == 1 && sg.cont)               // (issued with the load of done_at: one trip to the L2, not two)
      sl = __hip_atomic_load(sg.cont + (size_t)(sg.step & 1) * sg.count + member, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!known_live && sg.done[member].done_at <= sg.step) return;   // uniform over the workgroup: before any barrier (the one-launch kernel has just looked)
    slot2 = sl != 0;
  }
  if (ROLE == 3) {
    const int steps = sg.done[member].done_at <= sg.step ? sg.done[member].steps : sg.step;   // sg.step = steps enqueued
    if (steps & 1) Bb = B2b;
  }
  double* As = smem;
  double* Bs = smem + BK * LDS;
  double* Ct = smem;                // TM x (TM+1) transposed output tile (MIRROR), after the k loop
  const size_t mat = (size_t)member * (size_t)N * (size_t)N;
  const double* A = Ab + mat;
  const double* B = Bb + mat;
  const double* E = Eb ? Eb + mat : nullptr;
  double* C = ROLE == 5 ? Cb : Cb + mat;
  int by, bx;
  if (MIRROR) {
    if (sb > 0) {
      // XCD-aware order for one big matrix: workgroup i runs on XCD i % 8 (round-robin dispatch); give every XCD a
      // contiguous range of the tile sequence, and let the sequence walk 8x8-tile super-blocks of the upper triangle,
      // so that the ~128 tiles resident on an XCD share ~16 row/column strips in its private L2.
      const int per_xcd = grid_x / 8;
      const int L = (tile_x % 8) * per_xcd + tile_x / 8;
      int sbx, sby;
      tri_decode(L / 64, sbx, sby);
      by = sby * 8 + (L % 64) / 8;
      bx = sbx * 8 + (L % 64) % 8;
      if (sbx >= sb || bx < by || bx >= N / TM) return;   // whole workgroup leaves before any barrier
    } else {
      tri_decode(tile_x, bx, by);   // bx >= by
    }
  } else {
    const int nb = N / TM;
    by = tile_x / nb;
    bx = tile_x % nb;
  }
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wy = wave >> 1, wx = wave & 1;
  int row0 = by * TM, col0 = bx * TM;
  const int r16 = lane & 15, kk = lane >> 4;
  const int nbt = N / TM, ntiles = nbt * (nbt + 1) / 2;
  bool full = false;                // ROLE 1, second slot of a clean mega-lift: the full product M = R - R Y
  double gamma = 0.0;               // ROLE 2 / 4, second slot: C = E + gamma (M - A B)
  const double* Madd = nullptr;
  auto decide = [&]() {
    double mu, cm, cmc;
    int half;
    if (ROLE == 2) {
      mu = lg_reduce_decide(sg, member, ntiles, bx == 0 && by == 0, red);
      cm = red[13]; half = (int)red[14]; cmc = red[15];
    } else {
      const SignDevState& v = sg.st[(size_t)((sg.step & 1) ^ 1) * sg.count + member];
      mu = v.mu; cm = v.sched.cm; half = v.sched.half; cmc = v.sched.cmc;
    }
    alpha = cm > 0.0 ? -cm : -0.5 * mu * mu * mu;
    beta = cm > 0.0 ? 1.0 + cm : 1.5 * mu;
    if (CLEAN && half == 1) { alpha = -1.0; beta = 1.0; }
    if (CLEAN && half == 2) {       // A = Y, B = M, E = C = the old iterate (in the output buffer), + cmc M
      A = Bb + mat; B = B2b + mat; E = Cb + mat; Madd = B2b + mat;
      alpha = -cmc; beta = 1.0; gamma = cmc;
    }
  };
  // The decision only sets the epilogue's coefficients -- unless a clean mega-lift may change the operands.  Without one (ROLE 2) it runs
  // BEHIND the first operand loads: the slots' and the state's trips to the L2 overlap the operands' instead of preceding them.
  // (LATE = false: the one-launch kernel -- the operands live across the state machine cost 46 registers, and three of its workgroups must fit a CU)
  constexpr bool kLateDecision = ROLE == 2 && !CLEAN && LATE;
  if ((ROLE == 2 || ROLE == 4) && !kLateDecision) decide();
  if (CLEAN && ROLE == 1 && slot2) {      // A = R (the step before left it where the iterate would be), B = Y, C = M
    full = true;
    B = Cb + mat; E = Ab + mat; C = const_cast<double*>(B2b) + mat;
    alpha = -1.0; beta = 1.0;
  }

  double p0 = 0.0, p1 = 0.0;     // ROLE 1: tr Y, ||Y||_F^2; ROLE 2 / 4: ||S - S Y||_F^2 (this tile's share)
  // (a full product runs the tile and then its transpose: the same workgroup, operands swapped by symmetry of the ROLES, not of the result)
  const int npass = (CLEAN && full && by != bx) ? 2 : 1;
  for (int pass = 0; pass < npass; ++pass) {
  if (pass == 1) { const int t_ = by; by = bx; bx = t_; row0 = by * TM; col0 = bx * TM; }
  lg_v4f64 acc[NTW][NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = lg_v4f64{0.0, 0.0, 0.0, 0.0};

  // staging: tile[k][0..TM-1] = M[k0 + k][c0 .. c0+TM-1]; BK k-rows, 256/BK threads per row
  constexpr int TPR = 256 / BK;     // threads per k-row of the staged tile
  constexpr int PT = TM / TPR;      // doubles per thread per operand per k-tile (4 or 2)
  const int lk = tid / TPR, lc = (tid % TPR) * PT;
  // register prefetch two k-tiles ahead (an L2 miss costs several k-tiles of MFMA work); scalars, not arrays (scratch)
  double2 pa0, pa1 = {0, 0}, pb0, pb1 = {0, 0}, qa0, qa1 = {0, 0}, qb0, qb1 = {0, 0};
  const double2* ap = reinterpret_cast<const double2*>(A + (size_t)lk * N + row0 + lc);   // A symmetric: A[row][k] = A[k][row]
  const double2* bp = reinterpret_cast<const double2*>(B + (size_t)lk * N + col0 + lc);
  const size_t kstep = (size_t)BK * N / 2;
#define LG_LOAD(a0, a1, b0, b1)                                  \
  do {                                                           \
    a0 = ap[0]; b0 = bp[0];                                      \
    if constexpr (PT == 4) { a1 = ap[1]; b1 = bp[1]; }           \
  } while (0)
#define LG_STORE(a0, a1, b0, b1)                                 \
  do {                                                           \
    sa[0] = a0; sb2[0] = b0;                                     \
    if constexpr (PT == 4) { sa[1] = a1; sb2[1] = b1; }          \
  } while (0)
  LG_LOAD(pa0, pa1, pb0, pb1);
  ap += kstep; bp += kstep;
  if (N > BK) LG_LOAD(qa0, qa1, qb0, qb1);               // N is a multiple of BK; an odd number of k-tiles ends after a first half
  // the epilogue's E values travel with the operands (32 x 32 tiles: four per thread) instead of as a trip of their own behind the k loop
  constexpr bool kPreE = TM == 32;
  double epre[kPreE ? 4 : 1];
  if (kPreE && E) {
#pragma unroll
    for (int r = 0; r < 4; ++r) epre[r] = E[(size_t)(row0 + wy * WT + kk + 4 * r) * N + col0 + wx * WT + r16];
  }
  if (kLateDecision) decide();
  double2* sa = reinterpret_cast<double2*>(As + lk * LDS + lc);
  double2* sb2 = reinterpret_cast<double2*>(Bs + lk * LDS + lc);
#define LG_COMPUTE()                                                                                              \
  _Pragma("unroll") for (int ks = 0; ks < BK; ks += 4) {                                                       \
    double af[NTW], bf[NTW];                                                                                      \
    _Pragma("unroll") for (int t = 0; t < NTW; ++t) {                                                             \
      af[t] = As[(ks + kk) * LDS + wy * WT + t * 16 + r16];                                                       \
      bf[t] = Bs[(ks + kk) * LDS + wx * WT + t * 16 + r16];                                                       \
    }                                                                                                             \
    _Pragma("unroll") for (int i = 0; i < NTW; ++i)                                                               \
      _Pragma("unroll") for (int j = 0; j < NTW; ++j)                                                             \
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);                       \
  }
  for (int k0 = 0; k0 < N; k0 += 2 * BK) {
    __syncthreads();
    LG_STORE(pa0, pa1, pb0, pb1);
    __syncthreads();
    if (k0 + 2 * BK < N) { ap += kstep; bp += kstep; LG_LOAD(pa0, pa1, pb0, pb1); }
    LG_COMPUTE();
    if (k0 + BK >= N) break;                              // odd number of k-tiles (N a multiple of BK only): uniform
    __syncthreads();
    LG_STORE(qa0, qa1, qb0, qb1);
    __syncthreads();
    if (k0 + 3 * BK < N) { ap += kstep; bp += kstep; LG_LOAD(qa0, qa1, qb0, qb1); }
    LG_COMPUTE();
  }
#undef LG_COMPUTE
#undef LG_LOAD
#undef LG_STORE
  // epilogue: D[row = (l>>4) + 4*reg][col = l&15] per 16x16 tile.  The mirrored copy goes through LDS so that it is
  // stored row-wise too (a direct transposed store puts the 16 lanes of a fragment 8N bytes apart: one L2 channel).
  if (MIRROR) __syncthreads();   // everyone is done with As / Bs: Ct overlays them
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lrow = wy * WT + i * 16 + kk + 4 * r, lcol = wx * WT + j * 16 + r16;
        const int row = row0 + lrow, col = col0 + lcol;
        const size_t idx = (size_t)row * N + col;
        double c = alpha * acc[i][j][r];
        if (E) {
          const double ev = kPreE ? epre[r] : E[idx];
          c += beta * ev;
          if (ROLE == 2 || ROLE == 4) { const double d = ev - acc[i][j][r]; p0 += d * d; }
        }
        if (CLEAN && (ROLE == 2 || ROLE == 4) && Madd) c += gamma * Madd[idx];
        if (ROLE == 1) {
          p1 += acc[i][j][r] * acc[i][j][r];
          if (row == col) p0 += acc[i][j][r];
        }
        if (ROLE == 5) {
          if (row <= col && col < n_true) {
            if (!(fabs(c) <= 1.7976931348623157e308) && fail) atomicAdd(fail, 1);
            C[(size_t)col * (col + 1) / 2 + row] = row == col ? c : c * kSqrt2;
          }
        } else {
        if (!MIRROR || full || col >= row) C[idx] = c;        // diagonal tiles: the upper triangle decides (a full product: every entry is its own)
        if (MIRROR && !full) Ct[lcol * (TM + 1) + lrow] = c;
        }
      }
  if (MIRROR && !full && ROLE != 5) {
    __syncthreads();
    const int q = tid % TM;                           // original row   -> column of the mirrored tile
#pragma unroll 4
    for (int p = tid / TM; p < TM; p += 256 / TM)     // original column -> row of the mirrored tile
      if (by != bx || q < p) C[(size_t)(col0 + p) * N + row0 + q] = Ct[p * (TM + 1) + q];
  }
  }    // pass
  if ((ROLE == 1 && !full) || ROLE == 2 || ROLE == 4) {
    // this tile's partial sums -> its fixed slot (read by the next launch; the full product of a second slot leaves none: nothing reads them)
    const double w = (by == bx) ? 1.0 : 2.0;                 // an off-diagonal tile stands for its mirror image too
    p0 = wave_sum(p0);
    p1 = wave_sum(p1);
    __syncthreads();
    if (lane == 0) { red[wave] = p0; red[4 + wave] = p1; }
    __syncthreads();
    if (tid == 0) {
      const int slot = bx * (bx + 1) / 2 + by;
      const double s0 = (red[0] + red[1]) + (red[2] + red[3]);
      const double s1 = (red[4] + red[5]) + (red[6] + red[7]);
      if (ROLE == 1) {
        double* p = sg.p1 + ((size_t)member * ntiles + slot) * 2;
        p[0] = s0;
        p[1] = s1 * w;
      } else {
        sg.p2[((size_t)(sg.step & 1) * sg.count + member) * (size_t)ntiles + slot] = s0 * w;
      }
    }
  }
}

template <bool MIRROR, int TM, int BK, int ROLE, bool CLEAN = false>
__global__ __launch_bounds__(256) void lg_gemm_sym_kernel(int N, const double* __restrict__ Ab, const double* __restrict__ Bb,
                                                          double alpha, double beta, const double* __restrict__ Eb,
                                                          double* __restrict__ Cb, int sb, SignArgs sg, const double* __restrict__ B2b) {
  __shared__ double smem[LgGemmCfg<MIRROR, TM, BK>::SMEM];
  __shared__ double red[16];
  lg_gemm_sym_body<MIRROR, TM, BK, ROLE, CLEAN>(N, Ab, Bb, alpha, beta, Eb, Cb, sb, sg, B2b, (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x, smem, red);
}

// ---- a handful of mid-size blocks: the WHOLE sign iteration in one launch --------------------------------------------------
// A moment relaxation has a few blocks of 65 <= n <= 128 (PlanarHand_N=1: nine).  Their products are a few workgroups deep, so
// the ~95 dependent launches of a projection cost their launch latency (5.6 / 11.8 us each: 0.8 ms), not their flops.  Here one
// persistent workgroup per (member, output tile) runs every step: Y = S S, [barrier], T = 1.5 mu S - 0.5 mu^3 S Y, [barrier], ...
// and the final product.  The barrier is per MEMBER (its nb (nb + 1) / 2 workgroups): a monotone counter in global memory, an
// agent-scope release before the increment and an acquire after the wait (the workgroups of a member may sit on different
// XCDs, whose L2s are not coherent with each other).  Same tile bodies, same slots, same state machine as the launches: the
// results are bit-identical.  What bounds it: a phase is ~5 dependent trips to the device's coherence point (operands, stores
// complete, arrive, poll, the slots) of ~1.2 us each -- PlanarHand_N=1: 17.5 -> 15 us per step, projection 1.07 -> 0.80 ms.
// Measured and NOT faster (round 3): both operand panels resident in LDS with every load of a phase issued at once and the
// decision overlapped (0.84 ms); relaxed agent-scope atomic loads / stores for everything exchanged instead of the bulk
// release / acquire, the state carried per workgroup (0.80 ms) -- the trips, not the L2 maintenance, are the cost.  The grid must be co-resident (the host only takes this path for <= kClusterMaxWgs workgroups);
// a workgroup that waits longer than ~2 s gives up and raises the failure counter instead of hanging the device.
constexpr int LG_CS_ROWS = 32;       // row chunks of the column sums of |X| (lg_colsum_kernel; the one-launch kernel's prologue keeps the association)
constexpr int kClusterMaxWgs = 224;       // default of psd_lg_cluster_wgs
constexpr int kClusterXccStride = 1024;   // ints per table of d_xcc (the option's upper bound + slack)
struct ClusterArgs {
  double *S, *T, *Y, *X0, *M;
  unsigned* bar;            // per member, zeroed by lg_state_init_kernel
  int* xcc;                 // [member][tile]: the XCD every workgroup found itself on
  int* fail;
  int N, max_steps, count, force_agent;
  int spread;               // 1: workgroups in plain order (member = w / ntiles) -- a member too large for one XCD's CUs
  // FUSED (psd_lg_fuse): the one-launch kernel also does lg_prep_kernel's and lg_pack_steps_kernel's work -- its workgroups build X0, the column
  // sums and S between their first two barriers and the final product stores the svec: one launch per projection instead of three (the
  // prologue alone was 23 us of a 320 us chain on PlanarHand_N=1)
  int fused;
  double* colsum;           // [member][LG_CS_ROWS][N]
  unsigned* bar_other;      // the counters of the NEXT projection (two sets, alternating): zeroed by this one, since nobody zeroes this one's before it
};
// Several groups (different padded sizes N) in one launch: workgroup and member ranges per group.  A relaxation with blocks of 126 and
// of 252 (taha1a) would otherwise run its two one-launch groups one after the other, each a few workgroups deep and bound by its barriers.
constexpr int kClusterMaxGroups = 4;
struct ClusterMulti {
  int n;
  int wg_begin[kClusterMaxGroups + 1];      // multiples of 8 (workgroup w runs on XCD w % 8)
  int mem_begin[kClusterMaxGroups + 1];     // the prologue / epilogue launches run one workgroup (row) per member over all groups
  ClusterArgs ca[kClusterMaxGroups];
  SignArgs sg[kClusterMaxGroups];
  const double* in; const long long* boff; const int* bn; double* out; int* steps;      // FUSED: the plan's vectors
};
// local: every workgroup of the member sits on the SAME XCD (checked at run time, below).  Then their common L2 is the coherence
// point on the PRODUCER side: a workgroup signals once its own stores have completed (s_waitcnt vmcnt(0): the per-CU L1 writes
// through) -- no write-back of the XCD's L2 (buffer_wbl2), which is what the agent-scope release costs.  The waiter's side is the
// agent-scope invalidate, buffer_inv sc1.  (The first version issued buffer_inv sc0 -- workgroup scope -- to drop "only the CU's
// L1": outside threadgroup-split mode that does not invalidate the L1 at all.  A phase streams far more than an L1 through the CU,
// so stale operand lines were almost always evicted by then -- the bit-identity tests passed for a week -- until two groups in one
// launch (ClusterMulti) changed the footprint: 1 solve in 3 left the per-group launches' bits after 5 ... 50 iterations, and a
// two-rank taha1a run left the oracle's trajectory at 1e-2.  tools/dbg/merge_engine_check.py is the reproducer.)
// Otherwise: agent-scope release / acquire (buffer_wbl2 sc1 / buffer_inv sc1), correct across XCDs.
__device__ __forceinline__ bool lg_member_barrier(unsigned* bar, unsigned target, bool local) {
  if (local) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    if (!local) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int good = 1;
    long long spins = 0;
    while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1ll << 24)) { good = 0; break; }
    }
    if (local) asm volatile("buffer_inv sc1" ::: "memory");
    else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    ok = good;
  }
  __syncthreads();
  return ok != 0;
}
// 1-D grid: workgroup w runs on XCD w % 8 (round-robin dispatch: tools/ubench/xcc_probe.hip), so member m = (w % 8) + 8 j takes the
// workgroups w = (m % 8) + 8 (j ntiles + tile): all of a member's workgroups on one XCD.  Nothing RELIES on that: every workgroup
// publishes the XCD it really runs on (HW_REG_XCC_ID), and after a first (agent-scope) barrier each checks that its member's are equal
// -- only then the light barrier is used.  (Measured and rejected: workgroup-scope read-modify-writes on the counter, hoping they
// would be served by the shared L2 -- they are not coherent between CUs: the waiters time out.)
template <int TM, int BK, bool CLEAN>
__global__ __launch_bounds__(256) void lg_sign_cluster_kernel(ClusterMulti cm) {
  __shared__ double smem[LgGemmCfg<true, TM, BK>::SMEM];
  __shared__ double red[16];
  __shared__ int s_local;
  int gi = 0;
  while (gi + 1 < cm.n && (int)blockIdx.x >= cm.wg_begin[gi + 1]) ++gi;     // uniform: scalar loads from the kernel arguments
  const ClusterArgs ca = cm.ca[gi];      // by value: scalar registers, not a reload per use
  SignArgs sg = cm.sg[gi];
  const int N = ca.N;
  const int nbt = N / TM;
  const unsigned ntiles = (unsigned)(nbt * (nbt + 1) / 2);
  const int w = (int)blockIdx.x - cm.wg_begin[gi], slot = w >> 3;
  const int member = ca.spread ? w / (int)ntiles : (w & 7) + 8 * (slot / (int)ntiles);
  const int tile = ca.spread ? w % (int)ntiles : slot % (int)ntiles;
  if (member >= ca.count) return;               // the whole workgroup, before any barrier
  unsigned* bar = ca.bar + member;
  unsigned phase = 0;
  // FUSED prologue, first half (lg_prep_kernel's work spread over the member's workgroups): workgroup `tile` takes the row chunks tile,
  // tile + ntiles, ... of the LG_CS_ROWS chunks: the block's svec gathered into the dense X0, and per chunk and column the sum of |X0| over the
  // chunk's rows in row order -- lg_colsum_kernel's association, so the norm, the scale and S carry the bits of the other paths.
  const int fid = ca.fused ? sg.ids[member] : 0, fn = ca.fused ? cm.bn[fid] : 0;
  const int crow = (N + LG_CS_ROWS - 1) / LG_CS_ROWS;
  if (ca.fused) {
    const double* __restrict__ sv = cm.in + cm.boff[fid];
    double* __restrict__ X0 = ca.X0 + (size_t)member * N * N;
    double* __restrict__ cs = ca.colsum + (size_t)member * LG_CS_ROWS * N;
    for (int z = tile; z < LG_CS_ROWS; z += (int)ntiles) {
      const int r0 = z * crow, r1 = r0 + crow < N ? r0 + crow : N;
      for (int c = (int)threadIdx.x; c < N; c += 256) {
        double sum = 0.0;
        for (int r = r0; r < r1; ++r) {
          double v = 0.0;
          if (r < fn && c < fn) {
            const int lo = r < c ? r : c, hi = r < c ? c : r;
            v = sv[(long long)hi * (hi + 1) / 2 + lo];
            if (lo != hi) v *= kSqrt2Inv;
          }
          X0[(size_t)r * N + c] = v;
          sum += fabs(v);
        }
        cs[(size_t)z * N + c] = sum;
      }
    }
    if (tile == 0 && threadIdx.x == 0) ca.bar_other[member] = 0u;      // the next projection's counter (this one's was zeroed by the previous projection)
  }
  int xcc0;                                     // the XCD this workgroup started on
  {
    int x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    xcc0 = x & 15;
    if (threadIdx.x == 0) __hip_atomic_store(ca.xcc + (size_t)member * ntiles + tile, x & 15, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!lg_member_barrier(bar, ntiles * ++phase, false)) { if (threadIdx.x == 0 && ca.fail) atomicAdd(ca.fail, 1); return; }
    if (threadIdx.x == 0) {
      int same = ca.force_agent ? 0 : 1;
      for (unsigned q = 0; q < ntiles; ++q)
        same &= __hip_atomic_load(ca.xcc + (size_t)member * ntiles + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (x & 15);
      s_local = same;
    }
    __syncthreads();
  }
  const bool local = s_local != 0;
  if (ca.fused) {
    // second half: ||X0||_1 = the largest column sum (chunk sums added in chunk order: lg_scale_kernel), formed by every workgroup for itself;
    // S = X0 / ||X0||_1 on the workgroup's own rows (the values it stored above); workgroup 0 writes the member's fresh schedule state
    const double* __restrict__ cs = ca.colsum + (size_t)member * LG_CS_ROWS * N;
    double* __restrict__ X0 = ca.X0 + (size_t)member * N * N;
    double* __restrict__ S0 = ca.S + (size_t)member * N * N;
    double mx = 0.0;
    for (int c = (int)threadIdx.x; c < N; c += 256) {
      double v = 0.0;
      for (int z = 0; z < LG_CS_ROWS; ++z) v += cs[(size_t)z * N + c];
      mx = (v > mx || !(v == v)) ? v : mx;   // NaN propagates (flagged at the store of the projection)
    }
    double* redm = smem;                        // 256 doubles of the tile buffer (free until the first product)
    redm[threadIdx.x] = mx;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
      if ((int)threadIdx.x < k) { const double v = redm[threadIdx.x + k]; if (v > redm[threadIdx.x] || !(v == v)) redm[threadIdx.x] = v; }
      __syncthreads();
    }
    const double nrm = redm[0];
    __syncthreads();
    const double scale = nrm > 0.0 ? 1.0 / nrm : (nrm == 0.0 ? 0.0 : nrm);
    for (int z = tile; z < LG_CS_ROWS; z += (int)ntiles) {
      const int r0 = z * crow, r1 = r0 + crow < N ? r0 + crow : N;
      for (int c = (int)threadIdx.x; c < N; c += 256)
        for (int r = r0; r < r1; ++r) S0[(size_t)r * N + c] = X0[(size_t)r * N + c] * scale;
    }
    if (tile == 0 && threadIdx.x == 0) {
      if (member == 0) { sg.group[0] = sg.count; sg.group[1] = 0; }
      lg_fresh_state(sg, member, fid, fn, false);
    }
    if (!lg_member_barrier(bar, ntiles * ++phase, local)) { if (threadIdx.x == 0 && ca.fail) atomicAdd(ca.fail, 1); return; }
  }
  double* s = ca.S;
  double* t = ca.T;
  int step = 0;
  for (; step < ca.max_steps; ++step) {
    sg.step = step;
    // done_at is written by this member's writer workgroup during the second product of the step before: ordered by the barrier
    if (__hip_atomic_load(&sg.done[member].done_at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= step) break;
    lg_gemm_sym_body<true, TM, BK, 1, CLEAN, false>(N, s, s, 1.0, 0.0, nullptr, ca.Y, 0, sg, ca.M, member, tile, (int)ntiles, smem, red, true);
    if (!lg_member_barrier(bar, ntiles * ++phase, local)) { if (threadIdx.x == 0 && ca.fail) atomicAdd(ca.fail, 1); return; }
    lg_gemm_sym_body<true, TM, BK, 2, CLEAN, false>(N, s, ca.Y, 0.0, 0.0, s, t, 0, sg, ca.M, member, tile, (int)ntiles, smem, red, true);
    if (!lg_member_barrier(bar, ntiles * ++phase, local)) { if (threadIdx.x == 0 && ca.fail) atomicAdd(ca.fail, 1); return; }
    if (local) {
      // the light barrier stands on "same XCD", established once at the start: a workgroup that finds itself elsewhere (a preempted
      // queue restored on other hardware) raises the failure counter -- an error, not a silently stale operand
      int x;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
      if ((x & 15) != xcc0 && threadIdx.x == 0 && ca.fail) atomicAdd(ca.fail, 1);
    }
    double* u = s; s = t; t = u;
  }
  // P = 0.5 (X0 + X0 S_final); `s` holds the last iterate
  sg.step = step;
  if (ca.fused) {      // ... stored as the block's svec (ROLE 5), and the step count beside it: lg_pack_steps_kernel's work
    lg_gemm_sym_body<true, TM, BK, 5>(N, ca.X0, s, 0.5, 0.5, ca.X0, cm.out + cm.boff[fid], 0, sg, nullptr, member, tile, (int)ntiles, smem, red, false, fn, ca.fail);
    if (cm.steps && tile == 0 && threadIdx.x == 0)
      cm.steps[fid] = sg.done[member].done_at <= ca.max_steps ? sg.done[member].steps : ca.max_steps;
  } else {
    lg_gemm_sym_body<true, TM, BK, 0>(N, ca.X0, s, 0.5, 0.5, ca.X0, ca.Y, 0, sg, nullptr, member, tile, (int)ntiles, smem, red);
  }
}

// Group descriptors: member m of the group is block ids[m]; n = bn[id], svec offset boff[id].
// svec (upper triangle, column by column, sqrt2 on off-diagonals) -> dense N x N (padding pre-zeroed)
__global__ void lg_unpack_kernel(const double* __restrict__ src, const int* __restrict__ ids, const long long* __restrict__ boff,
                                 const int* __restrict__ bn, int N, double* __restrict__ dst) {
  const int id = ids[blockIdx.y];
  const int n = bn[id];
  const int len = n * (n + 1) / 2;
  const double* s = src + boff[id];
  double* d = dst + (size_t)blockIdx.y * N * N;
  for (int e = (int)(blockIdx.x * blockDim.x + threadIdx.x); e < len; e += (int)(gridDim.x * blockDim.x)) {
    int i, j;
    tri_decode(e, i, j);
    double v = s[e];
    if (i != j) v *= kSqrt2Inv;
    d[(size_t)j * N + i] = v;
    d[(size_t)i * N + j] = v;
  }
}
__global__ void lg_pack_kernel(const double* __restrict__ src, const int* __restrict__ ids, const long long* __restrict__ boff,
                               const int* __restrict__ bn, int N, double* __restrict__ dst, int* __restrict__ fail) {
  const int id = ids[blockIdx.y];
  const int n = bn[id];
  const int len = n * (n + 1) / 2;
  const double* s = src + (size_t)blockIdx.y * N * N;
  double* d = dst + boff[id];
  bool bad = false;
  for (int e = (int)(blockIdx.x * blockDim.x + threadIdx.x); e < len; e += (int)(gridDim.x * blockDim.x)) {
    int i, j;
    tri_decode(e, i, j);
    const double v = s[(size_t)j * N + i];
    bad |= !(fabs(v) <= 1.7976931348623157e308);
    d[e] = (i == j) ? v : v * kSqrt2;
  }
  if (bad && fail) atomicAdd(fail, 1);   // non-finite input: same counter as the QL sweep cap of the eigensolver kernels
}
// column sums of |X| (X symmetric: = row sums) in LG_CS_ROWS row chunks (blockIdx.z), then scale[m] = 1 / max_c colsum
// (0 for a zero block).  One thread per column over ALL rows takes 0.49 ms at N = 2048 (8 workgroups on 256 CUs).
__global__ __launch_bounds__(256) void lg_colsum_kernel(const double* __restrict__ Mb, int N, double* __restrict__ colsum) {
  const double* M = Mb + (size_t)blockIdx.y * N * N;
  const int col = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const int rows = (N + LG_CS_ROWS - 1) / LG_CS_ROWS, r0 = (int)blockIdx.z * rows, r1 = r0 + rows < N ? r0 + rows : N;
  if (col < N) {
    double s = 0.0;
    for (int r = r0; r < r1; ++r) s += fabs(M[(size_t)r * N + col]);
    colsum[((size_t)blockIdx.y * LG_CS_ROWS + blockIdx.z) * N + col] = s;
  }
}
__global__ __launch_bounds__(256) void lg_scale_kernel(const double* __restrict__ colsum, int N, double* __restrict__ scale) {
  __shared__ double red[256];
  double m = 0.0;
  for (int c = (int)threadIdx.x; c < N; c += 256) {
    double v = 0.0;
    for (int z = 0; z < LG_CS_ROWS; ++z) v += colsum[((size_t)blockIdx.x * LG_CS_ROWS + z) * N + c];   // fixed order
    m = (v > m || !(v == v)) ? v : m;   // NaN propagates (flagged by the pack kernel)
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const double v = red[threadIdx.x + s];
      if (v > red[threadIdx.x] || !(v == v)) red[threadIdx.x] = v;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) scale[blockIdx.x] = red[0] > 0.0 ? 1.0 / red[0] : (red[0] == 0.0 ? 0.0 : red[0]);
}
__global__ void lg_scaled_copy_kernel(const double* __restrict__ src, double* __restrict__ dst, size_t per_mat,
                                      const double* __restrict__ scale) {
  const double s = scale[blockIdx.y];
  const double* a = src + (size_t)blockIdx.y * per_mat;
  double* b = dst + (size_t)blockIdx.y * per_mat;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_mat; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i] * s;
}
// debug only: sum (A - B)^2 per matrix
__global__ __launch_bounds__(256) void lg_diff_kernel(const double* __restrict__ A, const double* __restrict__ B, size_t per_mat,
                                                      double* __restrict__ out) {
  __shared__ double red[256];
  const double* a = A + (size_t)blockIdx.x * per_mat;
  const double* b = B + (size_t)blockIdx.x * per_mat;
  double s = 0.0;
  for (size_t i = threadIdx.x; i < per_mat; i += 256) { const double d = a[i] - b[i]; s += d * d; }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) { if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

// The one-launch variant's prologue in ONE launch (one workgroup of 1024 threads per member) instead of six (zero fill, unpack,
// column sums, scale, scaled copy, state init: ~7 us each with their launch gaps -- a tenth of a PlanarHand projection): the svec is
// gathered straight into the dense X0 (every element written: no zero fill), the column sums of |X0| are formed in the association
// of lg_colsum_kernel / lg_scale_kernel (LG_CS_ROWS row chunks summed in chunk order: the same bits), S = X0 / ||X0||_1, and thread 0
// writes the member's fresh schedule state.  For the small N of that variant only (the matrix is walked by one workgroup).
__global__ __launch_bounds__(1024) void lg_prep_kernel(const double* __restrict__ src, const long long* __restrict__ boff,
                                                       const int* __restrict__ bn, ClusterMulti cm) {
  __shared__ double red[1024];
  int gi = 0;
  while (gi + 1 < cm.n && (int)blockIdx.x >= cm.mem_begin[gi + 1]) ++gi;
  const SignArgs& sa = cm.sg[gi];
  const int N = cm.ca[gi].N;
  double* __restrict__ X0b = cm.ca[gi].X0;
  double* __restrict__ Sb = cm.ca[gi].S;
  const int m = (int)blockIdx.x - cm.mem_begin[gi], tid = (int)threadIdx.x;
  const int id = sa.ids[m], n = bn[id];
  const double* sv = src + boff[id];
  double* X0 = X0b + (size_t)m * N * N;
  double* S = Sb + (size_t)m * N * N;
  for (int idx = tid; idx < N * N; idx += 1024) {
    const int r = idx / N, c = idx - r * N;
    double v = 0.0;
    if (r < n && c < n) {
      const int lo = r < c ? r : c, hi = r < c ? c : r;
      v = sv[(long long)hi * (hi + 1) / 2 + lo];
      if (lo != hi) v *= kSqrt2Inv;
    }
    X0[idx] = v;
  }
  __syncthreads();
  // column sums (lg_colsum_kernel: chunk z covers rows [z rows, (z + 1) rows), rows = ceil(N / LG_CS_ROWS); lg_scale_kernel: chunks in
  // order).  All 1024 threads: thread t forms the chunk sums z = t / N, t / N + 1024 / N, ... of column t % N into LDS; then one thread per
  // column adds its LG_CS_ROWS chunk sums in chunk order.
  extern __shared__ double lgp_cs[];           // LG_CS_ROWS x N
  const int rows = (N + LG_CS_ROWS - 1) / LG_CS_ROWS;
  if (tid < (1024 / N) * N) {
    const int c = tid % N;
    for (int z = tid / N; z < LG_CS_ROWS; z += 1024 / N) {
      const int r0 = z * rows, r1 = r0 + rows < N ? r0 + rows : N;
      double s = 0.0;
      for (int r = r0; r < r1; ++r) s += fabs(X0[(size_t)r * N + c]);
      lgp_cs[z * N + c] = s;
    }
  }
  __syncthreads();
  double mx = 0.0;
  for (int c = tid; c < N; c += 1024) {
    double v = 0.0;
    for (int z = 0; z < LG_CS_ROWS; ++z) v += lgp_cs[z * N + c];
    mx = (v > mx || !(v == v)) ? v : mx;   // NaN propagates (flagged by the pack kernel)
  }
  red[tid] = mx;
  __syncthreads();
  for (int k = 512; k > 0; k >>= 1) {
    if (tid < k) { const double v = red[tid + k]; if (v > red[tid] || !(v == v)) red[tid] = v; }
    __syncthreads();
  }
  const double scale = red[0] > 0.0 ? 1.0 / red[0] : (red[0] == 0.0 ? 0.0 : red[0]);
  for (int idx = tid; idx < N * N; idx += 1024) S[idx] = X0[idx] * scale;
  if (tid == 0) {
    if (m == 0) { sa.group[0] = sa.count; sa.group[1] = 0; }
    lg_fresh_state(sa, m, id, n);
  }
}
// lg_pack_kernel + lg_steps_out_kernel in one launch (the one-launch variant's epilogue)
__global__ void lg_pack_steps_kernel(const long long* __restrict__ boff, const int* __restrict__ bn, double* __restrict__ dst,
                                     int* __restrict__ fail, int* __restrict__ steps, ClusterMulti cm) {
  int gi = 0;
  while (gi + 1 < cm.n && (int)blockIdx.y >= cm.mem_begin[gi + 1]) ++gi;
  const SignArgs& sa = cm.sg[gi];             // sa.step = the steps enqueued (set by the host before this launch)
  const int N = cm.ca[gi].N;
  const int m = (int)blockIdx.y - cm.mem_begin[gi];
  const int id = sa.ids[m];
  const int n = bn[id];
  const int len = n * (n + 1) / 2;
  const double* s = cm.ca[gi].Y + (size_t)m * N * N;
  double* d = dst + boff[id];
  bool bad = false;
  for (int e = (int)(blockIdx.x * blockDim.x + threadIdx.x); e < len; e += (int)(gridDim.x * blockDim.x)) {
    int i, j;
    tri_decode(e, i, j);
    const double v = s[(size_t)j * N + i];
    bad |= !(fabs(v) <= 1.7976931348623157e308);
    d[e] = (i == j) ? v : v * kSqrt2;
  }
  if (bad && fail) atomicAdd(fail, 1);
  if (steps && blockIdx.x == 0 && threadIdx.x == 0) steps[id] = sa.done[m].done_at <= sa.step ? sa.done[m].steps : sa.step;
}

static int lg_pad(int n) { return (n + LG_TM - 1) / LG_TM * LG_TM; }

static bool lg_small_tiles(bool mirror, int N, int count, int tile_force = 0) {
  // 32 x 32 tiles when the 64 x 64 tiling would leave the chip under-filled (option psd_lg_tile = 32 | 64 forces one)
  const int nb64 = N / 64;
  const long long tiles64 = (long long)(mirror ? nb64 * (nb64 + 1) / 2 : nb64 * nb64) * count;
  // measured: better up to N ~ 3000, and down to N = 128 (a moment relaxation's handful of 65 <= n <= 128 blocks: the launch
  // is a few workgroups deep, and four times as many, four times smaller ones finish sooner -- PlanarHand_N=1: projection
  // 1.41 -> 1.02 ms)
  return tile_force ? tile_force == 32 : (mirror && N >= 128 && tiles64 < 1300);
}
// how a group's sign iteration is enqueued: tiles per member, decision as a launch of its own, everything in one launch
struct LgPath { int ntiles; bool decide_kernel, cluster; int cluster_wgs, spread; };

template <int ROLE>
static int lg_gemm_mirror(int N, int count, const double* A, const double* B, double alpha, double beta, const double* E, double* C,
                          hipStream_t st, const SignArgs& sa, const double* B2, int tile_force = 0) {
  const bool small_tiles = lg_small_tiles(true, N, count, tile_force);
  const int nb = small_tiles ? N / 32 : N / 64;
  constexpr bool kCanClean = ROLE == 1 || ROLE == 2 || ROLE == 4;
  if (count == 1 && nb >= 16) {
    const int sb = (nb + 7) / 8;                               // 8x8-tile super-blocks per direction
    if (sa.clean) { set_error("psd sign path: a clean mega-lift group on the super-block order (N = %d)", N); return CUADMM_ERR_INVALID; }
    if (small_tiles) hipLaunchKernelGGL((lg_gemm_sym_kernel<true, 32, 32, ROLE>), dim3(sb * (sb + 1) / 2 * 64), dim3(256), 0, st, N, A, B, alpha, beta, E, C, sb, sa, B2);
    else hipLaunchKernelGGL((lg_gemm_sym_kernel<true, 64, 16, ROLE>), dim3(sb * (sb + 1) / 2 * 64), dim3(256), 0, st, N, A, B, alpha, beta, E, C, sb, sa, B2);
  } else if (kCanClean && sa.clean) {
    if (small_tiles) hipLaunchKernelGGL((lg_gemm_sym_kernel<true, 32, 32, ROLE, kCanClean>), dim3(nb * (nb + 1) / 2, count), dim3(256), 0, st, N, A, B, alpha, beta, E, C, 0, sa, B2);
    else hipLaunchKernelGGL((lg_gemm_sym_kernel<true, 64, 16, ROLE, kCanClean>), dim3(nb * (nb + 1) / 2, count), dim3(256), 0, st, N, A, B, alpha, beta, E, C, 0, sa, B2);
  } else {
    if (small_tiles) hipLaunchKernelGGL((lg_gemm_sym_kernel<true, 32, 32, ROLE>), dim3(nb * (nb + 1) / 2, count), dim3(256), 0, st, N, A, B, alpha, beta, E, C, 0, sa, B2);
    else hipLaunchKernelGGL((lg_gemm_sym_kernel<true, 64, 16, ROLE>), dim3(nb * (nb + 1) / 2, count), dim3(256), 0, st, N, A, B, alpha, beta, E, C, 0, sa, B2);
  }
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

static LgPath lg_path(const PsdOptions& opt, int N, int cnt);
int large_gemm_sym(int N, const double* A, const double* B, double alpha, double beta, const double* E, double* C, hipStream_t st) {
  hipLaunchKernelGGL((lg_gemm_sym_kernel<false, 64, 16, 0>), dim3((N / 64) * (N / 64), 1), dim3(256), 0, st, N, A, B, alpha, beta, E, C, 0,
                     SignArgs{}, (const double*)nullptr);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

// one thread per member: fresh schedule state (version 0)
__global__ void lg_state_init_kernel(SignArgs sa, const int* __restrict__ ids, const int* __restrict__ bn) {
  const int m = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (m == 0) { sa.group[0] = sa.count; sa.group[1] = 0; }
  if (m >= sa.count) return;
  lg_fresh_state(sa, m, ids[m], bn[ids[m]]);
}
__global__ void lg_steps_out_kernel(SignArgs sa, const int* __restrict__ ids, int* __restrict__ steps) {
  const int m = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (m >= sa.count) return;
  steps[ids[m]] = sa.done[m].done_at <= sa.step ? sa.done[m].steps : sa.step;
}

// ------------------------------------------------------------------------------------------
// Planner: groups by padded size, chunks of bounded workspace
// ------------------------------------------------------------------------------------------
int SignPsd::build(const int* blk, const std::vector<int>& members) {
  release();
  if (members.empty()) return CUADMM_OK;
  size_t ws_cap = (size_t)8 << 30;   // bytes of workspace (4 matrices per member); 288 GB of HBM make this generous
  ws_cap = (size_t)std::max(1, opt.sign_ws_mb) << 20;
  std::map<int, std::vector<int>> by_pad;
  for (int k : members) by_pad[lg_pad(blk[k])].push_back(k);
  std::vector<int> ids;
  size_t max_elems = 0, max_cols = 0;
  int max_count = 0;
  for (auto& kv : by_pad) {
    int N = kv.first;
    const size_t per = (size_t)N * N * sizeof(double) * 4;
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>(ws_cap / per, 65535));
    // A group on the 32 x 32 tiles only needs a multiple of 32: when every member fits N - 32 the padding shrinks (n = 2000:
    // 2016 instead of 2048 is 4.6 % fewer flops per product; n = 861: 864 instead of 896 is 10 %)
    {
      int nmax = 0;
      for (int k : kv.second) nmax = std::max(nmax, blk[k]);
      const int cnt = (int)std::min<size_t>((size_t)chunk, kv.second.size());
      const int tf = opt.lg_tile;
      if (opt.lg_pad32 && nmax <= N - 32 && lg_small_tiles(true, N, cnt, tf) && lg_small_tiles(true, N - 32, cnt, tf) &&
          lg_small_tiles(true, N - 32, (int)(kv.second.size() % (size_t)chunk ? kv.second.size() % (size_t)chunk : cnt), tf))
        N -= 32;
    }
    for (size_t b = 0; b < kv.second.size(); b += (size_t)chunk) {
      Group g;
      g.N = N;
      g.begin = (int)ids.size();
      g.count = (int)std::min<size_t>((size_t)chunk, kv.second.size() - b);
      for (int i = 0; i < g.count; ++i) ids.push_back(kv.second[b + i]);
      max_elems = std::max(max_elems, (size_t)g.count * N * N);
      max_cols = std::max(max_cols, (size_t)g.count * N * LG_CS_ROWS);
      max_count = std::max(max_count, g.count);
      groups.push_back(g);
    }
  }
  // One-launch groups of DIFFERENT padded sizes (a relaxation with blocks of 126 and of 252: taha1a) share ONE launch
  // (ClusterMulti), each with a workspace of its own, while all of them together stay within the co-residency bound of such a
  // launch.  Everything else runs one group after the other in the shared region.  (First attempt: side streams.  A plain side
  // stream landed on the caller's hardware queue -- no overlap at all; a high-priority one overlapped, and every small kernel of the
  // caller's stream took 40 us instead of 5.)
  auto part_of = [](const Group& g) { return (size_t)g.count * 2 * (size_t)(g.N / 32) * (size_t)(g.N / 32 + 1) / 2; };
  size_t max_part = 0, side_elems = 0, side_part = 0;
  int side_members = 0, n_merged = 0, wgs = 0;
  // per-XCD load of the merged launch: member q of a non-spread group sits on XCD q % 8 (every group's first workgroup is a
  // multiple of 8), a spread group's workgroups go round the XCDs.  lg_path bounds one group's share of an XCD (96 workgroups, the
  // XCD's CUs hold ~128); several groups must respect it TOGETHER, or a later group's workgroups wait for CUs while the resident
  // ones spin in lg_member_barrier (no deadlock -- barriers are per member -- but serial, and the ~2 s give-up budget runs)
  int xcd_load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (Group& g : groups) {
    const LgPath p = lg_path(opt, g.N, g.count);
    if (!(opt.lg_merge != 0 && !opt.debug && p.cluster && g.N <= 512 && n_merged < kClusterMaxGroups &&
          wgs + g.count * p.ntiles <= std::max(opt.lg_cluster_wgs, 8)))   // the workgroups that stay (the grid's others leave at once)
      continue;
    int add[8];
    bool fits = true;
    for (int x = 0; x < 8; ++x) {
      add[x] = p.spread ? (g.count * p.ntiles + 7 - x) / 8 : ((g.count > x ? (g.count - x + 7) / 8 : 0) * p.ntiles);
      fits = fits && xcd_load[x] + add[x] <= 96;
    }
    if (!fits) continue;
    for (int x = 0; x < 8; ++x) xcd_load[x] += add[x];
    wgs += g.count * p.ntiles;
    g.merged = true;
    g.slot = 1 + n_merged++;
  }
  if (n_merged < 2)                             // nothing to merge: the plain path
    for (Group& g : groups) { g.merged = false; g.slot = 0; }
  if (n_merged < 2) n_merged = 0;
  max_elems = 0; max_count = 0;                 // the shared region: the groups of the caller's stream only
  for (const Group& g : groups)
    if (!g.merged) {
      max_elems = std::max(max_elems, (size_t)g.count * g.N * g.N);
      max_count = std::max(max_count, g.count);
      max_part = std::max(max_part, part_of(g));
    }
  for (Group& g : groups)
    if (g.merged) {                             // behind the shared region
      g.ws_off = max_elems + side_elems;
      g.mem_off = max_count + side_members;
      g.part_off = max_part + side_part;
      side_elems += (size_t)g.count * g.N * g.N;
      side_members += g.count;
      side_part += part_of(g);
    }
  const size_t elems = std::max<size_t>(max_elems + side_elems, 1), members_cap = std::max<size_t>((size_t)max_count + (size_t)side_members, 1);
  CUADMM_HIP_TRY(hipMalloc(&d_ids, sizeof(int) * ids.size()));
  { int rc_ = staged_h2d(d_ids, ids.data(), sizeof(int) * ids.size()); if (rc_) return rc_; }
  CUADMM_HIP_TRY(hipMalloc(&X0, sizeof(double) * elems));
  CUADMM_HIP_TRY(hipMalloc(&S, sizeof(double) * elems));
  CUADMM_HIP_TRY(hipMalloc(&Y, sizeof(double) * elems));
  CUADMM_HIP_TRY(hipMalloc(&T, sizeof(double) * elems));
  {
    bool any_clean = false;
    for (const Group& g : groups) any_clean = any_clean || (opt.lg_clean != 0 && g.N <= clean_max_n);
    if (any_clean) {
      CUADMM_HIP_TRY(hipMalloc(&Mw, sizeof(double) * elems));
      CUADMM_HIP_TRY(hipMalloc(&d_cont, sizeof(int) * 2 * members_cap));
    }
  }
  CUADMM_HIP_TRY(hipMalloc(&colsum, sizeof(double) * std::max<size_t>(max_cols, 1)));
  CUADMM_HIP_TRY(hipMalloc(&scale, sizeof(double) * std::max<size_t>(members_cap, 1)));
  CUADMM_HIP_TRY(hipMalloc(&d_state, sizeof(SignDevState) * 2 * members_cap));
  CUADMM_HIP_TRY(hipMalloc(&d_done, sizeof(SignDone) * members_cap));
  CUADMM_HIP_TRY(hipMalloc(&d_group, sizeof(int) * 2 * (size_t)(1 + n_merged)));
  CUADMM_HIP_TRY(hipMalloc(&d_bar, sizeof(unsigned) * 3 * members_cap));      // sets 0 / 1: fused projections, alternating; set 2: the others (zeroed at their start)
  CUADMM_HIP_TRY(hipMemset(d_bar, 0, sizeof(unsigned) * 3 * members_cap));
  bar_stride = members_cap;
  {
    size_t cs_total = 0;
    for (Group& g : groups) { g.cs_off = cs_total; cs_total += (size_t)g.count * (size_t)g.N * LG_CS_ROWS; }
    CUADMM_HIP_TRY(hipMalloc(&colsum_f, sizeof(double) * std::max<size_t>(cs_total, 1)));
  }
  CUADMM_HIP_TRY(hipMalloc(&d_xcc, sizeof(int) * (size_t)(kClusterXccStride) * (size_t)(1 + n_merged)));
  CUADMM_HIP_TRY(hipHostMalloc(&h_group, sizeof(int) * 2, hipHostMallocDefault));
  part_half = max_part + side_part;
  CUADMM_HIP_TRY(hipMalloc(&d_part, sizeof(double) * 2 * std::max<size_t>(part_half, 1)));   // p1 | p2 (two parities)
  return CUADMM_OK;
}

void SignPsd::release() {
  if (graph_exec) { hipError_t e = hipGraphExecDestroy(graph_exec); (void)e; graph_exec = nullptr; }
  for (void* p : {(void*)d_ids, (void*)X0, (void*)S, (void*)Y, (void*)T, (void*)colsum, (void*)scale, (void*)d_state, (void*)d_part, (void*)d_done, (void*)d_bar, (void*)d_xcc, (void*)Mw, (void*)d_cont, (void*)colsum_f})
    if (p) { hipError_t e = hipFree(p); (void)e; }
  if (d_group) { hipError_t e = hipFree(d_group); (void)e; d_group = nullptr; }
  if (h_group) { hipError_t e = hipHostFree(h_group); (void)e; h_group = nullptr; }
  d_ids = nullptr; d_state = nullptr; d_part = nullptr; d_done = nullptr; d_bar = nullptr; d_xcc = nullptr;
  X0 = S = Y = T = colsum = scale = nullptr;
  Mw = nullptr; d_cont = nullptr; colsum_f = nullptr;
  groups.clear();
}

// out = svec(P_+(smat(in))) for every member block; boff / bn are the plan's device arrays (all blocks).  Asynchronous.
// The fixed schedule is ~95 dependent launches per group (1.0 ms for a handful of N = 128 blocks, ~10 us per dependent
// kernel).  Replaying them from a hipGraph was measured and does NOT help (1.835 vs 1.814 ms per PlanarHand projection):
// the cost is the device-side drain/flush between dependent kernels, not host launch overhead.  Kept behind
// CUADMM_PSD_GRAPH=1 for re-measurement on other ROCm versions.
int SignPsd::project(const double* in, double* out, const long long* boff, const int* bn, int* d_fail, hipStream_t st) {
  const bool use_graph = opt.graph == 1 && !opt.debug;
  if (!use_graph || !allow_graph || st == nullptr || groups.empty()) return project_launch(in, out, boff, bn, d_fail, st);
  if (graph_exec && (g_in != in || g_out != out || g_boff != boff || g_bn != bn || g_fail != d_fail)) {
    hipError_t e = hipGraphExecDestroy(graph_exec); (void)e;
    graph_exec = nullptr;
  }
  if (!graph_exec) {
    hipGraph_t graph = nullptr;
    CUADMM_HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc = project_launch(in, out, boff, bn, d_fail, st);
    hipError_t e = hipStreamEndCapture(st, &graph);
    if (rc) { if (graph) { hipError_t e2 = hipGraphDestroy(graph); (void)e2; } return rc; }
    if (e != hipSuccess) { set_error("psd sign path: graph capture failed: %s", hipGetErrorString(e)); return CUADMM_ERR_INVALID; }
    e = hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0);
    { hipError_t e2 = hipGraphDestroy(graph); (void)e2; }
    if (e != hipSuccess) { graph_exec = nullptr; set_error("psd sign path: graph instantiate failed: %s", hipGetErrorString(e)); return CUADMM_ERR_INVALID; }
    g_in = in; g_out = out; g_boff = boff; g_bn = bn; g_fail = d_fail;
  }
  CUADMM_HIP_TRY(hipGraphLaunch(graph_exec, st));
  return CUADMM_OK;
}

static LgPath lg_path(const PsdOptions& opt, int N, int cnt) {
  LgPath p{};
  // The decision of a step sums the tiles' slots: inside every workgroup of the S Y product when they are few, as a launch
  // of its own when that would cost more than a launch (option psd_lg_decide = 1 | 2 forces one).
  const int nbt = lg_small_tiles(true, N, cnt, opt.lg_tile) ? N / 32 : N / 64;
  p.ntiles = nbt * (nbt + 1) / 2;
  p.decide_kernel = opt.lg_decide ? opt.lg_decide == 2 : p.ntiles > 300;
  // a handful of mid-size blocks: every step and the final product in ONE launch (lg_sign_cluster_kernel)
  p.cluster = opt.lg_cluster != 0 && !p.decide_kernel && lg_small_tiles(true, N, cnt, opt.lg_tile) && (long long)p.ntiles * cnt <= std::max(opt.lg_cluster_wgs, 8);
  // one XCD has 32 CUs x 4 workgroups of this kernel: a member's workgroups go to ONE XCD only while everything mapped there stays
  // co-resident with room to spare; else plain order over the whole chip (agent-scope barriers)
  p.spread = ((cnt + 7) / 8) * p.ntiles > 96 ? 1 : 0;
  p.cluster_wgs = p.spread ? cnt * p.ntiles : 8 * ((cnt + 7) / 8) * p.ntiles;
  return p;
}

int SignPsd::project_launch(const double* in, double* out, const long long* boff, const int* bn, int* d_fail, hipStream_t st) {
  const bool sync_ok = opt.sign_sync != 0;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (st) { hipError_t e = hipStreamIsCapturing(st, &cap); (void)e; }
  const bool poll = sync_ok && cap == hipStreamCaptureStatusNone;
  const int max_steps = opt.sign_maxsteps > 0 ? opt.sign_maxsteps : SignSched::kCap;
  int rc;
  // the merged one-launch groups: three launches for all of them (prologue, sign iteration, epilogue); then the rest, group by group
  ClusterMulti cm{};
  for (const Group& g : groups)
    if (g.merged) cluster_add(cm, g, d_fail, max_steps);
  if (cm.n > 0 && (rc = cluster_run(cm, true, in, out, boff, bn, d_fail, st))) return rc;
  for (Group& g : groups)
    if (!g.merged && (rc = launch_group(g, in, out, boff, bn, d_fail, st, poll))) return rc;
  return CUADMM_OK;
}

// the group's share of the workspace and of the schedule's state as arguments of the one-launch kernels
void SignPsd::cluster_add(ClusterMulti& cm, const Group& g, int* d_fail, int max_steps) const {
  const int i = cm.n++;
  const LgPath path = lg_path(opt, g.N, g.count);
  SignArgs& sa = cm.sg[i];
  sa = SignArgs{};
  sa.st = static_cast<SignDevState*>(d_state) + 2 * (size_t)g.mem_off;
  sa.done = static_cast<SignDone*>(d_done) + g.mem_off;
  sa.p1 = d_part + g.part_off;
  sa.p2 = d_part + part_half + g.part_off;
  sa.group = d_group + 2 * g.slot;
  sa.hint = g.N <= hint_max_n ? d_hint : nullptr;
  sa.ids = d_ids + g.begin;
  sa.count = g.count;
  sa.step = 0;
  // fused prologue / epilogue (psd_lg_fuse; not under a graph capture's replay -- the counter set alternates per projection -- and only where the
  // one-launch prologue applies at all, N <= 512): this projection counts in set bar_par and zeroes the other one
  const bool fused = opt.lg_fuse != 0 && opt.graph != 1 && !opt.debug && g.N <= 512;
  // (the groups that run one after the other share the region at offset 0 and with it the counters: ONE alternation for all of them)
  int& bar_par = g.merged ? g.bar_par : shared_bar_par;
  const int par = fused ? bar_par : 2;          // (a projection that is not fused zeroes its counters itself, in a set of its own: it must not dirty the alternating ones)
  if (fused) bar_par ^= 1;
  sa.bar = d_bar + (size_t)par * bar_stride + g.mem_off;
  sa.clean = (Mw && opt.lg_clean != 0 && g.N <= clean_max_n) ? 1 : 0;
  sa.cont = sa.clean ? d_cont + 2 * (size_t)g.mem_off : nullptr;
  sa.cap = max_steps;
  // psd_lg_cluster = 2: agent-scope barriers always (A/B, tests)
  cm.ca[i] = ClusterArgs{S + g.ws_off, T + g.ws_off, Y + g.ws_off, X0 + g.ws_off, sa.clean ? Mw + g.ws_off : nullptr, sa.bar,
                         d_xcc + (size_t)kClusterXccStride * (size_t)g.slot, d_fail, g.N, max_steps, g.count, opt.lg_cluster == 2 ? 1 : 0, path.spread,
                         fused ? 1 : 0, colsum_f + g.cs_off, d_bar + (size_t)(fused ? par ^ 1 : 2) * bar_stride + g.mem_off};
  cm.wg_begin[i + 1] = cm.wg_begin[i] + (path.cluster_wgs + 7) / 8 * 8;
  cm.mem_begin[i + 1] = cm.mem_begin[i] + g.count;
}

// prologue (optional: the caller has filled X0, S and the state itself), every step and the final product, epilogue: three launches
int SignPsd::cluster_run(ClusterMulti& cm, bool prologue, const double* in, double* out, const long long* boff, const int* bn, int* d_fail, hipStream_t st) {
  int maxN = 0;
  for (int i = 0; i < cm.n; ++i) maxN = std::max(maxN, cm.ca[i].N);
  const bool fused = cm.n > 0 && cm.ca[0].fused != 0;
  for (int i = 0; i < cm.n; ++i)      // cluster_add chose the counter sets by it: all groups of a launch alike, and never without the prologue being this launch's to do
    if ((cm.ca[i].fused != 0) != fused || (fused && !prologue)) { set_error("psd sign path: fused and plain one-launch groups in one launch"); return CUADMM_ERR_INVALID; }
  cm.in = in; cm.boff = boff; cm.bn = bn; cm.out = out; cm.steps = d_steps;
  if (prologue && !fused) {
    const size_t lds_prep = sizeof(double) * LG_CS_ROWS * (size_t)maxN;
    static LdsCapOnce once;                      // the cap is lifted to the kernel's maximum, once per device
    if (lds_prep > 48 * 1024) CUADMM_HIP_TRY(once(reinterpret_cast<const void*>(lg_prep_kernel)));
    hipLaunchKernelGGL(lg_prep_kernel, dim3(cm.mem_begin[cm.n]), dim3(1024), lds_prep, st, in, boff, bn, cm);
  }
  bool any_clean = false;                      // (the instantiation without the clean mega-lift's second slot when no group takes one: not a register more)
  for (int i = 0; i < cm.n; ++i) any_clean = any_clean || cm.sg[i].clean != 0;
  if (any_clean) hipLaunchKernelGGL((lg_sign_cluster_kernel<32, 32, true>), dim3(cm.wg_begin[cm.n]), dim3(256), 0, st, cm);
  else hipLaunchKernelGGL((lg_sign_cluster_kernel<32, 32, false>), dim3(cm.wg_begin[cm.n]), dim3(256), 0, st, cm);
  for (int i = 0; i < cm.n; ++i) cm.sg[i].step = cm.ca[i].max_steps;          // the steps enqueued
  const unsigned gx = (unsigned)std::min<size_t>(((size_t)maxN * maxN / 2 + 255) / 256, 1024);
  if (!fused) hipLaunchKernelGGL(lg_pack_steps_kernel, dim3(gx, cm.mem_begin[cm.n]), dim3(256), 0, st, boff, bn, out, d_fail, d_steps, cm);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

int SignPsd::launch_group(Group& g, const double* in, double* out, const long long* boff, const int* bn, int* d_fail, hipStream_t st, bool poll) {
  const int debug = opt.debug;
  const int max_steps = opt.sign_maxsteps > 0 ? opt.sign_maxsteps : SignSched::kCap;
  {
    const int N = g.N, cnt = g.count;
    const size_t per = (size_t)N * N;
    const int* ids = d_ids + g.begin;
    const unsigned gx = (unsigned)std::min<size_t>((per / 2 + 255) / 256, 1024);
    const auto t0 = std::chrono::steady_clock::now();
    const LgPath path = lg_path(opt, N, cnt);
    const int ntiles = path.ntiles;
    const bool decide_kernel = path.decide_kernel, cluster = path.cluster;
    // this group's share of the workspace (psd_large.h: Group)
    double* const X0 = this->X0 + g.ws_off;
    double* const S = this->S + g.ws_off;
    double* const Y = this->Y + g.ws_off;
    double* const T = this->T + g.ws_off;
    double* const scale = this->scale + g.mem_off;
    int* const d_group = this->d_group + 2 * g.slot;
    unsigned* const d_bar = this->d_bar + 2 * bar_stride + g.mem_off;      // (set 2: cluster_add's choice for a projection that is not fused)
    SignArgs sa{};
    sa.st = static_cast<SignDevState*>(d_state) + 2 * (size_t)g.mem_off;
    sa.done = static_cast<SignDone*>(d_done) + g.mem_off;
    sa.p1 = d_part + g.part_off;
    sa.p2 = d_part + part_half + g.part_off;
    sa.group = d_group;
    sa.hint = g.N <= hint_max_n ? d_hint : nullptr;
    sa.ids = ids;
    sa.count = cnt;
    sa.step = 0;
    sa.bar = d_bar;
    sa.clean = (Mw && opt.lg_clean != 0 && g.N <= clean_max_n) ? 1 : 0;
    sa.cont = sa.clean ? d_cont + 2 * (size_t)g.mem_off : nullptr;
    sa.cap = max_steps;
    double* const M = sa.clean ? Mw + g.ws_off : nullptr;
    ClusterMulti cm{};
    if (cluster) cluster_add(cm, g, d_fail, max_steps);
    if (!(cluster && N <= 512)) {              // else: the one-launch variant's own prologue (cluster_run)
      CUADMM_HIP_TRY(hipMemsetAsync(X0, 0, sizeof(double) * per * (size_t)cnt, st));
      hipLaunchKernelGGL(lg_unpack_kernel, dim3(gx, cnt), dim3(256), 0, st, in, ids, boff, bn, N, X0);
      hipLaunchKernelGGL(lg_colsum_kernel, dim3((N + 255) / 256, cnt, LG_CS_ROWS), dim3(256), 0, st, X0, N, colsum);
      hipLaunchKernelGGL(lg_scale_kernel, dim3(cnt), dim3(256), 0, st, colsum, N, scale);
      hipLaunchKernelGGL(lg_scaled_copy_kernel, dim3(gx, cnt), dim3(256), 0, st, X0, S, per, scale);
      hipLaunchKernelGGL(lg_state_init_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, sa, ids, bn);
    }
    CUADMM_HIP_TRY(hipGetLastError());
    double* s = S;
    double* t = T;
    int rc;
    // Steps are enqueued in chunks; between chunks the host polls "members not finished" (8 bytes, one stream
    // synchronisation of ~20 us) instead of enqueueing the worst case kCap: a launch for a finished group costs ~5 us,
    // i.e. 0.5 ms per projection when the blocks stop after 16 of 64 steps.  The first chunk is the previous projection's
    // step count + 2 (ADMM iterates change slowly), so there is normally exactly one poll.  Inside a graph capture (or
    // with CUADMM_PSD_SIGN_SYNC=0) the whole cap is enqueued and nothing synchronises.
    int enq = 0;
    int chunk = poll ? std::min(max_steps, g.pred > 0 ? g.pred + 2 : 24) : max_steps;
    if (cluster) {
      if ((rc = cluster_run(cm, N <= 512, in, out, boff, bn, d_fail, st))) return rc;
      enq = max_steps;
    }
    while (enq < max_steps) {
      for (int it = 0; it < chunk && enq < max_steps; ++it, ++enq) {
        // Y = S*S ; [decision] ; T = 1.5 mu S - 0.5 mu^3 S*Y ; finished members return at once
        sa.step = enq;
        if ((rc = lg_gemm_mirror<1>(N, cnt, s, s, 1.0, 0.0, nullptr, Y, st, sa, M, opt.lg_tile))) return rc;
        if (decide_kernel) {
          hipLaunchKernelGGL(lg_decide_kernel, dim3(cnt), dim3(256), 0, st, sa, ntiles);
          if ((rc = lg_gemm_mirror<4>(N, cnt, s, Y, 0.0, 0.0, s, t, st, sa, M, opt.lg_tile))) return rc;
        } else {
          if ((rc = lg_gemm_mirror<2>(N, cnt, s, Y, 0.0, 0.0, s, t, st, sa, M, opt.lg_tile))) return rc;
        }
        std::swap(s, t);
      }
      if (!poll) break;
      CUADMM_HIP_TRY(hipMemcpyAsync(h_group, d_group, sizeof(int) * 2, hipMemcpyDeviceToHost, st));
      CUADMM_HIP_TRY(hipStreamSynchronize(st));
      if (h_group[0] <= 0) { g.pred = h_group[1]; break; }
      chunk = 6;
    }
    // P = 0.5 * (X0 + X0 * S_final); S_final is in S after an even number of steps, in T after an odd one
    sa.step = enq;
    if (!cluster && (rc = lg_gemm_mirror<3>(N, cnt, X0, S, 0.5, 0.5, X0, Y, st, sa, T, opt.lg_tile))) return rc;
    if (!cluster) {
      hipLaunchKernelGGL(lg_pack_kernel, dim3(gx, cnt), dim3(256), 0, st, Y, ids, boff, bn, N, out, d_fail);
      if (d_steps) hipLaunchKernelGGL(lg_steps_out_kernel, dim3((cnt + 255) / 256), dim3(256), 0, st, sa, ids, d_steps);
    }
    CUADMM_HIP_TRY(hipGetLastError());
    if (debug) {
      CUADMM_HIP_TRY(hipStreamSynchronize(st));
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      std::vector<SignDone> hs((size_t)cnt);
      { int rc_ = staged_d2h(hs.data(), sa.done, sizeof(SignDone) * (size_t)cnt, st); if (rc_) return rc_; }
      double steps = 0;
      int smax = 0;
      for (const SignDone& x : hs) { steps += x.steps; smax = std::max(smax, x.steps); }
      const int nb = N / LG_TM;
      const double flops = (2.0 * steps + cnt) * 2.0 * (double)N * LG_TM * LG_TM * (nb * (nb + 1) / 2);
      fprintf(stderr, "[psd debug] sign path: %d blocks padded to N=%d: steps mean %.1f max %d (%d enqueued, decision %s), %.2f ms, %.1f TFLOP/s fp64 MFMA (upper-triangle tiles)\n",
              cnt, N, steps / cnt, smax, enq, decide_kernel ? "kernel" : "inline", ms, flops / ms * 1e-9);
    }
  }
  return CUADMM_OK;
}

}  // namespace cuadmm
