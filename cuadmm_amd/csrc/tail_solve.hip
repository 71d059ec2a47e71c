// Dense tail of the A A^T solve on the GPU.
//
// The reference runs both triangular sweeps of CHOLMOD's LDL^T on the host every iteration (cholesky_cpu.h:146-155,
// solver.cu:487-500).  With a fill-reducing ordering the last k columns of L are an almost dense triangle L22 that
// holds most of nnz(L), and sweeping it from host DRAM is what bounds the iteration on moment-relaxation problems
// (PlanarHand_N=1: 13.5 M nonzeros, 16 ms per solve, 89 % of the iteration).  Here
//     x2 = L22^-T D2^-1 L22^-1 z2
// runs on the GPU as two triangular GEMVs with W = inv(L22) resident in HBM (W and W^T, both row-major, so that
// both GEMVs are row dot products: coalesced, no atomics, bit-reproducible across ranks).  W is computed once at
// init by recursive doubling on the fp64 matrix cores:
//     inv([L11 0; L21 L22]) = [W11 0; -W22 L21 W11, W22],
// 64 x 64 diagonal blocks by forward substitution, then block sizes 64, 128, 256, ... with two batched GEMMs per
// level (all groups of a level in one launch).  The host keeps the sparse leading columns (aat_ldlt.cpp).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "common.h"
#include "device_util.h"
#include "tail_solve.h"
#include "wave_reduce.h"

namespace cuadmm {

typedef double ts_v4f64 __attribute__((ext_vector_type(4)));
constexpr int TS_BK = 16, TS_TM = 64;

// C = alpha * A * B, general row-major operands with leading dimensions, batched (blockIdx.y) with element strides.
// M, N, Kd multiples of 64 / 64 / 16.  Same tiling as lg_gemm_sym_kernel (psd_large.hip); A is staged transposed.
__global__ __launch_bounds__(256) void ts_gemm_nn_kernel(int M, int N, int Kd, double alpha,
                                                         const double* __restrict__ Ab, long long lda, long long sA,
                                                         const double* __restrict__ Bb, long long ldb, long long sB,
                                                         double* __restrict__ Cb, long long ldc, long long sC) {
  constexpr int LDS = TS_TM + 16, WT = TS_TM / 2, NTW = WT / 16;
  __shared__ double As[TS_BK * LDS];
  __shared__ double Bs[TS_BK * LDS];
  const double* A = Ab + (size_t)blockIdx.y * sA;
  const double* B = Bb + (size_t)blockIdx.y * sB;
  double* C = Cb + (size_t)blockIdx.y * sC;
  const int tn = N / TS_TM;
  const int by = (int)blockIdx.x / tn, bx = (int)blockIdx.x % tn;
  (void)M;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wy = wave >> 1, wx = wave & 1;
  const int row0 = by * TS_TM, col0 = bx * TS_TM;
  const int r16 = lane & 15, kk = lane >> 4;
  ts_v4f64 acc[NTW][NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = ts_v4f64{0.0, 0.0, 0.0, 0.0};
  // A tile: 64 rows x 16 k; thread t: row t/4, k (t%4)*4 .. +3 (contiguous in memory), stored transposed As[k][row]
  const int ar = tid >> 2, ak = (tid & 3) * 4;
  const double2* ap = reinterpret_cast<const double2*>(A + (size_t)(row0 + ar) * lda + ak);
  // B tile: 16 k x 64 cols; thread t: k t/16, 4 cols at (t%16)*4
  const int lk = tid >> 4, lc = (tid & 15) * 4;
  const double2* bp = reinterpret_cast<const double2*>(B + (size_t)lk * ldb + col0 + lc);
  const size_t bstep = (size_t)TS_BK * ldb / 2;
  double2 pa0 = ap[0], pa1 = ap[1], pb0 = bp[0], pb1 = bp[1];
  double2* sb = reinterpret_cast<double2*>(Bs + lk * LDS + lc);
  for (int k0 = 0; k0 < Kd; k0 += TS_BK) {
    __syncthreads();
    As[(ak + 0) * LDS + ar] = pa0.x; As[(ak + 1) * LDS + ar] = pa0.y;
    As[(ak + 2) * LDS + ar] = pa1.x; As[(ak + 3) * LDS + ar] = pa1.y;
    sb[0] = pb0; sb[1] = pb1;
    __syncthreads();
    if (k0 + TS_BK < Kd) { ap += TS_BK / 2; bp += bstep; pa0 = ap[0]; pa1 = ap[1]; pb0 = bp[0]; pb1 = bp[1]; }
#pragma unroll
    for (int ks = 0; ks < TS_BK; ks += 4) {
      double af[NTW], bf[NTW];
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        af[t] = As[(ks + kk) * LDS + wy * WT + t * 16 + r16];
        bf[t] = Bs[(ks + kk) * LDS + wx * WT + t * 16 + r16];
      }
#pragma unroll
      for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + wy * WT + i * 16 + kk + 4 * r, col = col0 + wx * WT + j * 16 + r16;
        C[(size_t)row * ldc + col] = alpha * acc[i][j][r];
      }
}

// W_bb = inv(L_bb) for every 64 x 64 diagonal block (unit lower triangular): thread c builds column c by forward substitution
__global__ __launch_bounds__(64) void ts_diag_inverse_kernel(const double* __restrict__ L, double* __restrict__ W, long long ld) {
  __shared__ double Ls[64 * 65];
  const size_t base = (size_t)blockIdx.x * 64 * ld + (size_t)blockIdx.x * 64;
  const int c = (int)threadIdx.x;
  for (int r = 0; r < 64; ++r) Ls[r * 65 + c] = L[base + (size_t)r * ld + c];
  __syncthreads();
  double w[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) w[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    double s = (i == c) ? 1.0 : 0.0;
#pragma unroll
    for (int j = 0; j < i; ++j) s -= Ls[i * 65 + j] * w[j];   // w[j] = 0 for j < c
    w[i] = (i >= c) ? s : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 64; ++i) Ls[i * 65 + c] = w[i];
  __syncthreads();
  for (int r = 0; r < 64; ++r) W[base + (size_t)r * ld + c] = Ls[r * 65 + c];
}

__global__ __launch_bounds__(256) void ts_transpose_kernel(const double* __restrict__ src, double* __restrict__ dst, long long ld) {
  __shared__ double t[32][33];
  const int tx = (int)threadIdx.x & 31, ty = (int)threadIdx.x >> 5;
  const size_t r0 = (size_t)blockIdx.y * 32, c0 = (size_t)blockIdx.x * 32;
  for (int r = ty; r < 32; r += 8) t[r][tx] = src[(r0 + r) * ld + c0 + tx];
  __syncthreads();
  for (int r = ty; r < 32; r += 8) dst[(c0 + r) * ld + r0 + tx] = t[tx][r];
}

// out[i] = scale[i] * sum_{c in range(i)} Mx[i][c] * in[c];  LOWER: c <= i,  else c >= i.  One wavefront per row.
// Sharded solve (ranks split the rows of W, TailSolve::shard_*): the LOWER pass leaves 0 in the rows r = K - 1 - i outside [r_begin, r_end).
template <bool LOWER>
__global__ __launch_bounds__(256) void ts_tri_gemv_kernel(const double* __restrict__ Mx, long long ld, int K,
                                                          const double* __restrict__ in, const double* __restrict__ scale,
                                                          double* __restrict__ out, int r_begin = 0, int r_end = 1 << 30) {
  const int lane = (int)threadIdx.x & 63;
  // long rows first (they decide the tail of the launch)
  const int slot = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
  if (slot >= K) return;
  const int i = LOWER ? K - 1 - slot : slot;
  if (LOWER && (slot < r_begin || slot >= r_end)) { if (lane == 0) out[i] = 0.0; return; }
  const int lo = LOWER ? 0 : (i & ~1), hi = LOWER ? i + 1 : K;   // even start keeps the double2 loads aligned
  const double* row = Mx + (size_t)i * ld;
  double s = 0.0;
  int c = lo + 2 * lane;
  for (; c + 1 < hi; c += 128) {
    const double2 mv = *reinterpret_cast<const double2*>(row + c);
    const double2 xv = *reinterpret_cast<const double2*>(in + c);
    s += mv.x * xv.x + mv.y * xv.y;      // entries outside the triangle are exact zeros
  }
  if (c < hi) s += row[c] * in[c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) out[i] = scale ? s * scale[i] : s;
}


// refinement (option tail_refine): out_i = a_i - b_i - sum_{c < i} M[i][c] b_c (LOWER: M = L, unit diagonal implicit) or sum_{c > i} (M = L^T)
template <bool LOWER>
__global__ __launch_bounds__(256) void ts_tri_resid_kernel(const double* __restrict__ Mx, long long ld, int K, const double* __restrict__ a,
                                                           const double* __restrict__ b, double* __restrict__ out) {
  const int lane = (int)threadIdx.x & 63;
  const int i = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
  if (i >= K) return;
  const int lo = LOWER ? 0 : i + 1, hi = LOWER ? i : K;
  const double* row = Mx + (size_t)i * ld;
  double s = 0.0;
  for (int c = lo + lane; c < hi; c += 64) s += row[c] * b[c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) out[i] = (a[i] - b[i]) - s;
}
// pinv[perm[a]] = a: where a producer puts entry i of z so that the one-pass kernels read z in the factor's pivoting order WITHOUT a gather
__global__ void ts_perm_inverse_kernel(const int* __restrict__ perm, int* __restrict__ pinv, int K) {
  const int a = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (a < K) pinv[perm[a]] = a;
}
// dst[a] = src[perm[a]] (GATHER) or dst[perm[a]] = src[a]: in and out of the factor's pivoting order
template <bool GATHER>
__global__ void ts_perm_kernel(double* __restrict__ dst, const double* __restrict__ src, const int* __restrict__ perm, int K) {
  const int a = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (a < K) { if (GATHER) dst[a] = src[perm[a]]; else dst[perm[a]] = src[a]; }
}
// y += x (SCALE: then y *= d)
template <bool SCALE>
__global__ void ts_axpy1_kernel(double* __restrict__ y, const double* __restrict__ x, const double* __restrict__ d, int K) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i < K) { const double v = y[i] + x[i]; y[i] = SCALE ? v * d[i] : v; }
}

// x = W^T diag(dinv) W z in ONE pass over W.  The two triangular GEMVs above read W and then W^T: 8 K^2 bytes per solve, and they
// are what bounds the y-solve of a moment relaxation (PlanarHand_N=1, K = 17 152: 215 + 194 us at 5.5 - 6 TB/s).  Row i of W
// serves both products -- u_i = W_i z, then x += (dinv_i u_i) W_i^T -- so a workgroup that keeps the row in REGISTERS between
// the two uses needs it from memory once: 1024 threads, thread t holds NC columns of the current
// row(s) and of its partial x; z sits in LDS; one block reduction per row group (double-buffered: one barrier).  Workgroup g
// takes rows K-1-g, K-1-g-G, ... (long rows first, interleaved: equal shares of the triangle); the G partial vectors are summed
// in workgroup order by ts_onepass_reduce_kernel.  Deterministic; half the HBM traffic.
// Round 6, 16-BYTE LOADS: thread t holds column PAIRS -- every wavefront owns a contiguous segment of 64 NC columns, pair slot p of it is
// one coalesced 1 KB access (global_load_dwordx4) -- instead of single columns through 8-byte loads, which this chip serves at 0.54 - 0.70 of
// the 16-byte rate (MI355X_MICROARCH.md; measured here: 4.4 TB/s at K = 10 240 with dwordx2).  Every load is unconditional and straight-line
// (a pair above the diagonal reads the row's first pair and is zeroed by a select; a group beyond the range reads row 0 and is multiplied
// by zero).  PF: the rows of the NEXT group are in flight while the current group is reduced and applied (two register buffers; the
// compiler's counter for the current group's data, `s_waitcnt vmcnt(N)`, leaves the next group's loads outstanding across the barrier).
// The association of the sums follows the new column ownership: results differ from rounds 3 - 5 in the last bits, deterministically.
// A wavefront's 64 NC columns relate to row i in one of three ways, and which one is wave-uniform (seg0 = first column of the segment, in scalar
// registers): INSIDE the triangle (seg0 + 64 NC - 1 <= i: plain loads from one address register and immediates), ABOVE the diagonal (seg0 > i:
// nothing to read, nothing to multiply -- the wavefront only takes part in the row's barrier), or CROSSING it (one wavefront per row: per-pair
// predicates; a pair above the diagonal reads the row's first pair and is zeroed by a select).  Half of all (wavefront, row) pairs are ABOVE:
// before round 6's last version every one of them issued its loads, selects and 4 NC multiply-adds on the port the busy wavefronts share.
template <int NC>
__device__ __forceinline__ int ts_seg_class(int seg0, int i) { return seg0 + 64 * NC - 1 <= i ? 0 : (seg0 > i ? 2 : 1); }
template <int NC, int RB>
struct TsRows {
  static_assert(NC % 2 == 0, "column pairs");
  double2 w[RB][NC / 2];
  __device__ __forceinline__ void load(const double* __restrict__ W, long long ld, int K, int col0, int seg0, int r0, int r_end) {
#pragma unroll
    for (int q = 0; q < RB; ++q) {
      // (a group beyond the range re-reads the range's LAST row -- always a row this rank holds, keep_shard -- and is multiplied by zero)
      const int i = K - 1 - (r0 + q < r_end ? r0 + q : r_end - 1);
      const double* row = W + (long long)i * ld;
      const int cls = ts_seg_class<NC>(seg0, i);
      if (cls == 0) {
#pragma unroll
        for (int p = 0; p < NC / 2; ++p) w[q][p] = *reinterpret_cast<const double2*>(row + col0 + 128 * p);
      } else if (cls == 1) {
#pragma unroll
        for (int p = 0; p < NC / 2; ++p) {
          const int col = col0 + 128 * p;          // the pair (col, col + 1) is inside the triangle iff col <= i (its second half iff col < i)
          const double2 v = *reinterpret_cast<const double2*>(row + (col <= i ? col : 0));
          w[q][p].x = col <= i ? v.x : 0.0;        // entries above the diagonal count as exact zeros and are not read
          w[q][p].y = col < i ? v.y : 0.0;
        }
      }                                            // (ABOVE: the registers keep whatever they held; ts_rows_apply does not touch them)
    }
  }
};
// error-free transformations for the compensated variant of u_i = W_i z (option tail_dd, an experiment: NOTEBOOK.md "Round 6", "pivots near the
// regularisation"): (hi, lo) <- (hi, lo) + a b with the rounding errors of the product and of the sum collected in lo
__device__ __forceinline__ void dd_fma_acc(double& hi, double& lo, double a, double b) {
#pragma clang fp contract(off)               // the transformations are exact only with every operation rounded on its own
  const double pr = a * b, pe = __fma_rn(a, b, -pr);
  const double s = hi + pr, bb = s - hi;
  lo += ((hi - (s - bb)) + (pr - bb)) + pe;
  hi = s;
}
__device__ __forceinline__ void dd_add(double& hi, double& lo, double h2, double l2) {
#pragma clang fp contract(off)
  const double s = hi + h2, bb = s - hi;
  lo += ((hi - (s - bb)) + (h2 - bb)) + l2;
  hi = s;
}
// the sum of the sixteen wavefronts' parts, in every lane: lane l takes part l & 15, four butterfly steps on the DPP crossbar inside each row of
// 16 lanes -- the tree (((0+1)+(2+3))+((4+5)+(6+7))) + (((8+9)+(10+11))+((12+13)+(14+15))) of rounds 3 - 6, bit for bit (every step adds a pair
// that the old expression added, in one order or the other), for ONE LDS read per wavefront instead of eight 16-byte broadcasts and fifteen adds
__device__ __forceinline__ double ts_sum16(const double* rr, int lane) {
  double v = rr[lane & 15];
  v += sw_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
  v += sw_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  v += sw_dpp<0x141>(v);   // row_half_mirror
  v += sw_dpp<0x140>(v);   // row_mirror
  return v;
}
// ZREG: the thread's NC entries of z live in registers (zr) instead of LDS.  They are the same for every row -- the LDS copy only existed for the
// registers' sake -- and reading them back cost 64 NC bytes of LDS traffic per row and wavefront: with the sixteen wavefronts of a CU behind one
// LDS port, a third of a row's time (tools/ubench/tri_stream.hip, MODE 3: the tick stamps of wavefront 0).
template <int NC, int RB, bool DD, int NW = 16, bool TIGHT = false, bool ZREG = false>
__device__ __forceinline__ void ts_rows_apply(const TsRows<NC, RB>& R, const double* zs, const double2 (&zr)[ZREG ? NC / 2 : 1], double (*red)[RB][DD ? 32 : 16],
                                              int it, int lane, int wave, int seg0, int K, int r0, int r_end, const double* __restrict__ dinv,
                                              double2 (&xa)[NC / 2]) {
  static_assert(NW == 16 || NW == 8, "wavefronts per workgroup");
  static_assert(!DD || NW == 16, "the compensated variant exists for 1024 threads only");
  double part[RB], dv[RB];
  bool busy[RB];                             // wave-uniform: the wavefront's segment reaches into row q (ts_seg_class: INSIDE or CROSSING)
#pragma unroll
  for (int q = 0; q < RB; ++q) {
    const int i = r0 + q < r_end ? K - 1 - (r0 + q) : -1;
    busy[q] = seg0 <= (i >= 0 ? i : K - r_end);     // (a row beyond the range: the class of the row that was loaded in its place)
    dv[q] = i >= 0 && busy[q] ? dinv[i] : 0.0;      // the pivot travels (a scalar load) while the dot product is formed, not after the barrier
  }
#pragma unroll
  for (int q = 0; q < RB; ++q) {
    part[q] = 0.0;
    if constexpr (DD) {
      double lo = 0.0;
      if (busy[q]) {
#pragma unroll
        for (int p = 0; p < NC / 2; ++p) {
          const double2 zz = *reinterpret_cast<const double2*>(zs + 128 * p);
          dd_fma_acc(part[q], lo, R.w[q][p].x, zz.x);
          dd_fma_acc(part[q], lo, R.w[q][p].y, zz.y);
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) dd_add(part[q], lo, __shfl_xor(part[q], o, 64), __shfl_xor(lo, o, 64));
      }
      if (lane == 0) { red[it & 1][q][wave] = part[q]; red[it & 1][q][16 + wave] = lo; }
    } else {
      if (busy[q]) {
#pragma unroll
        for (int p = 0; p < NC / 2; ++p) {
          double2 zz;
          if constexpr (ZREG) zz = zr[p]; else zz = *reinterpret_cast<const double2*>(zs + 128 * p);
          part[q] += R.w[q][p].x * zz.x;
          part[q] += R.w[q][p].y * zz.y;
          // (TIGHT: at the register budget's edge the scheduler otherwise reads every z pair of the row up front -- 2 NC registers -- and spills)
          if (TIGHT && (p & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        part[q] = wave_sum(part[q]);
      }
      if (lane == 0) red[it & 1][q][wave] = part[q];
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < RB; ++q) {
    if (!busy[q]) continue;                  // every entry of the row in this wavefront's columns is an exact zero
    const double* rr = red[it & 1][q];
    double u;
    if constexpr (DD) {
      double hi = rr[0], lo = rr[16];
#pragma unroll
      for (int w = 1; w < 16; ++w) dd_add(hi, lo, rr[w], rr[16 + w]);
      u = hi + lo;
    } else if constexpr (NW == 8) {
      u = ((rr[0] + rr[1]) + (rr[2] + rr[3])) + ((rr[4] + rr[5]) + (rr[6] + rr[7]));
    } else {
      u = ts_sum16(rr, lane);
    }
    const double vq = u * dv[q];
#pragma unroll
    for (int p = 0; p < NC / 2; ++p) { xa[p].x += vq * R.w[q][p].x; xa[p].y += vq * R.w[q][p].y; }
  }
}
// NT threads: 1024 (sixteen wavefronts, <= 128 VGPRs each) or 512 -- "fewer, fatter threads", option tail_fat, measured slower and off.
// WHAT BOUNDS THIS KERNEL (round 6, tools/ubench/tri_stream.hip + the kernel traces profiles/r06_tail_onepass_study.txt): at K = 9 216 it takes
// 76.8 us for 340 MB whatever is changed inside it -- rows per barrier (1 / 2), groups in flight (0 ... 3), the walk, z in LDS or in registers,
// the above-diagonal wavefronts idle or busy, 8- or 16-wavefront workgroups: 76.8 - 78.5 us every time.  A FLAT 16-byte-per-lane read of the same
// bytes takes 60 - 62 us on the same box (5.5 TB/s at this footprint; 6.3 TB/s at 205 MB, where the Infinity Cache holds the operand), and the
// kernel adds what a flat read does not have: z and the pivot order gathered at the start (two dependent trips before the first dot product),
// 256 partial vectors written at the end (19 MB) and the ramp of 256 workgroups that all start on their longest row.  The tick stamps of
// wavefront 0 (MODE 3 of the micro-benchmark) show it never waits for its own row -- it waits at the barrier for the wavefront whose data is
// last: the kernel runs at the rate the memory system delivers this access pattern, 79 % of the measured flat-read ceiling, 55 % of 8 TB/s.
// D: row groups in flight BEYOND the one being applied (a ring of D + 1 register buffers; 0: load, apply, load ...).  With one group ahead
// the wait at the top of a group still sees a whole memory round trip minus the ~0.3 us a group takes to apply: a row per round trip,
// whatever its length -- the short rows of the triangle's tip are latency-bound (tools/ubench/tri_stream.hip).
// ORDER 1: a workgroup walks its rows alternately from the long and from the short end (ORDER 2, the default: odd workgroups start at the short
// end -- 77.7 -> 72.3 / 78.2 -> 75.2 us at K = 9 216, 45.5 -> 44.3 at 7 168 in the kernel trace), so the chip streams the same mix of long and short
// rows from the first microsecond to the last (ORDER 0, rounds 3 - 6: longest first -- every workgroup reaches the tip at the same time and
// the bytes in flight collapse together).  The partial sums associate in the walk's order: the last bits differ between orders, deterministically.
template <int NC, int RB, int D, bool DD = false, int NT = 1024, int ORDER = 0, bool ZREG = false>
__global__ __launch_bounds__(NT) void ts_onepass_kernel(const double* __restrict__ W, long long ld, int K, const double* __restrict__ z,
                                                        const double* __restrict__ dinv, double* __restrict__ P, int r_begin, int r_end,
                                                        const int* __restrict__ perm) {
  constexpr int NW = NT / 64;
  constexpr bool TIGHT = NT == 1024 && !ZREG && NC * RB * (D + 1) + NC >= 40;      // doubles of row data and accumulators per thread: 80 of the 128 registers
  extern __shared__ __attribute__((aligned(16))) double ts_zs[];          // z, K doubles (zero beyond K up to NT NC)
  __shared__ double red[2][RB][DD ? 32 : 16];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col0 = wave * (64 * NC) + 2 * lane;
  const int seg0 = __builtin_amdgcn_readfirstlane(wave) * (64 * NC);      // first column of the wavefront's segment, wave-uniform (ts_seg_class)
  const int G = (int)gridDim.x, g = (int)blockIdx.x, step = G * RB;
  // rows r = K - 1 - i in [r_begin, r_end): the whole triangle, or this rank's share of it (TailSolve::shard_*)
  const int first = r_begin + g * RB;
  const int count = first < r_end ? (r_end - first + step - 1) / step : 0;      // row groups of this workgroup
  if (count == 0) {                          // an empty share (tail_shard_bound) or more workgroups than row groups: zeros, and no row is touched
#pragma unroll
    for (int p = 0; p < NC / 2; ++p) { const int col = col0 + 128 * p; if (col < K) *reinterpret_cast<double2*>(P + (size_t)g * K + col) = make_double2(0.0, 0.0); }
    return;
  }
  // first row of the j-th group of the walk (beyond the last group: the last group again -- rows this rank holds; ts_rows_apply is not called for it)
  auto group_row = [&](int j) -> int {
    j = j < count ? j : count - 1;
    if (ORDER >= 1) { const int h = j >> 1; j = (((j & 1) != 0) != (ORDER == 2 && (g & 1) != 0)) ? count - 1 - h : h; }      // ORDER 2: odd workgroups start at the short end
    return first + j * step;
  };
  TsRows<NC, RB> buf[D + 1];
#pragma unroll
  for (int d = 0; d < (D > 0 ? D : 1); ++d) buf[d].load(W, ld, K, col0, seg0, group_row(d), r_end);      // the first groups travel while z is staged
  double2 zr[ZREG ? NC / 2 : 1];
  double2 xa[NC / 2];
#pragma unroll
  for (int p = 0; p < NC / 2; ++p) xa[p] = make_double2(0.0, 0.0);
  if constexpr (ZREG) {                      // (perm: the factor's pivoting order, TailSolve::perm_d)
#pragma unroll
    for (int p = 0; p < NC / 2; ++p) {
      const int col = col0 + 128 * p;
      zr[p] = col >= K ? make_double2(0.0, 0.0) : perm ? make_double2(z[perm[col]], z[perm[col + 1]]) : *reinterpret_cast<const double2*>(z + col);
    }
  } else {
    for (int c = tid; c < NT * NC; c += NT) ts_zs[c] = c < K ? z[perm ? perm[c] : c] : 0.0;
    __syncthreads();
  }
  const double* zs = ts_zs + col0;
  int it = 0;
  if constexpr (D == 0) {                    // one group in flight (the registers of a second one would spill)
    for (int j = 0; j < count; ++j) {
      if (j) buf[0].load(W, ld, K, col0, seg0, group_row(j), r_end);
      ts_rows_apply<NC, RB, DD, NW, TIGHT, ZREG>(buf[0], zs, zr, red, it++, lane, wave, seg0, K, group_row(j), r_end, dinv, xa);
    }
  } else {
    int j = 0;
    while (j < count) {
#pragma unroll
      for (int s = 0; s <= D; ++s) {
        if (j < count) {
          buf[(s + D) % (D + 1)].load(W, ld, K, col0, seg0, group_row(j + D), r_end);
          ts_rows_apply<NC, RB, DD, NW, TIGHT, ZREG>(buf[s], zs, zr, red, it++, lane, wave, seg0, K, group_row(j), r_end, dinv, xa);
          ++j;
        }
      }
    }
  }
#pragma unroll
  for (int p = 0; p < NC / 2; ++p) {
    const int col = col0 + 128 * p;
    if (col < K) *reinterpret_cast<double2*>(P + (size_t)g * K + col) = xa[p];       // K is a multiple of 64: col + 1 < K too
  }
}
// The one-pass product for tails BEYOND one workgroup's reach (18 432 < K <= 32 768: PushT_N=30 27 136, PushBox N=50 30 720, PlanarHand
// N=10 32 768).  A row of W no longer fits one CU's registers beside its accumulators (3 x 256 KB at K = 32 768 against 512 KB of
// registers + 160 KB of LDS), so Q = 4 workgroups share a row: member q owns the column segments (wave * Q + q) * 64 NC ... (interleaved:
// equal shares of the triangle), forms its part of u_i = W_i z, publishes it in a slot of its own (one double per row and member:
// a single atomic word, valid as soon as it is not the sentinel NaN -- no flag, no fence) and reads the three others; the sum is taken
// in member order, so it does not depend on who arrives first.  RB rows share one exchange (lane q of the first wavefront handles row
// q: the polls of a round are in flight together), z sits in LDS words only the owning thread touches, and either two workgroups share
// a CU (64 VGPRs) or one keeps four rows in flight (128), so the ~2 us exchange is amortised or covered.  The four members of a group
// are workgroups g, g + 8, g + 16, g + 24: one XCD (round-robin dispatch), one L2.  A group's members lie within 32 consecutive
// workgroup ids, so the dispatcher (in id order) always has whole groups resident; a member that waits ~seconds gives up and poisons the
// result with NaN instead of hanging.  W is read once: 4 K^2 bytes instead of the 8 K^2 of the two triangular GEMVs.
// Failure protocol (HIP does not PROMISE co-residency): a member that gives up raises *fail and returns the canonical NaN; every
// other waiter polls *fail between its spins and gives up with it, a launch that starts with *fail != 0 writes NaN and returns at
// once (no further 4 s budgets), and a member never PUBLISHES the sentinel: a NaN of its own -- hardware propagates payloads, so a
// poisoned right-hand side could carry the sentinel's bits back in -- is canonicalised before the store.  The engine sees the NaN in
// the iteration's scalars, reads the counter, clears it and moves the handle to the two-GEMV path (engine.hip, TailSolve::take_failure).
constexpr unsigned long long TS_SENTINEL = 0x7ff8dead5eed0001ull;      // a quiet NaN that no sum is allowed to publish
constexpr unsigned long long TS_CANON_NAN = 0x7ff8000000000000ull;
__global__ void ts_fill_u64_kernel(unsigned long long* p, size_t n, unsigned long long v) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
template <int NC, int Q, int RB, int OCC = 8, bool PF = false>
__global__ __launch_bounds__(1024, OCC) void ts_onepass_group_kernel(const double* __restrict__ W, long long ld, int K, const double* __restrict__ z,
                                                                   const double* __restrict__ dinv, double* __restrict__ P,
                                                                   unsigned long long* __restrict__ part, int* __restrict__ fail, int r_begin, int r_end,
                                                                   const int* __restrict__ perm, int order) {
  extern __shared__ double ts_zs[];          // z of this thread's own columns: word (c * 1024 + tid); nobody else reads it
  __shared__ double red[2][RB][16];
  __shared__ double ush[2][RB];
  static_assert(Q == 2 || Q == 4 || Q == 8, "members per row");
  constexpr int LQ = Q == 8 ? 3 : Q == 4 ? 2 : 1;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = (int)blockIdx.x;
  const int member = (g >> 3) & (Q - 1), group = (g & 7) + 8 * (g >> (3 + LQ));
  const int G = (int)gridDim.x / Q;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // round 6: column PAIRS per thread (global_load_dwordx4; the 8-byte loads of rounds 4 - 5 run at 0.54 - 0.70 of that rate): pair slot p of the
  // wavefront's segment is columns col0 + 128 p, + 1
  static_assert(NC % 2 == 0, "column pairs");
  constexpr int NP = NC / 2;
  const int col0 = (wave * Q + member) * (64 * NC) + 2 * lane;
  if (fail && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {   // an earlier exchange was lost: no more waiting
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int col = col0 + 128 * p;
      const double nan = __longlong_as_double((long long)TS_CANON_NAN);
      if (col < K) *reinterpret_cast<double2*>(P + (size_t)group * K + col) = make_double2(nan, nan);
    }
    return;
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int col = col0 + 128 * p;
    const double2 zz = col >= K ? make_double2(0.0, 0.0) : perm ? make_double2(z[perm[col]], z[perm[col + 1]])
                                                                   : *reinterpret_cast<const double2*>(z + col);      // K is a multiple of 64: col + 1 < K too
    ts_zs[(2 * p) * 1024 + tid] = zz.x; ts_zs[(2 * p + 1) * 1024 + tid] = zz.y;
  }
  double2 xa[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) xa[p] = make_double2(0.0, 0.0);
  // the group's row groups, walked longest first (order 0) or alternately from the long and the short end (order 1, ts_onepass_kernel): the
  // members of a group share `group` and `count`, so they walk the same sequence and meet in the same exchange slots
  const int first = r_begin + group * RB, stride = G * RB;
  const int count = first < r_end ? (r_end - first + stride - 1) / stride : 0;
  auto row_of = [&](int it) -> int {
    const int jw = order ? ((((it & 1) != 0) != (order == 2 && (group & 1) != 0)) ? count - 1 - (it >> 1) : (it >> 1)) : it;
    return first + jw * stride;
  };
  // the rows of one group into registers (w), and everything behind the loads (process)
  auto load_group = [&](double2 (&w)[RB][NP], const int r0) __attribute__((always_inline)) {
    const int seg0 = (wave_u * Q + member) * (64 * NC);   // this wavefront's segment: uniform
#pragma unroll
    for (int q = 0; q < RB; ++q) {
      const int i = r0 + q < r_end ? K - 1 - (r0 + q) : -1;            // < 0: no such row (all-zero contribution)
      const double* row = W + (long long)(i < 0 ? K - 1 - r_begin : i) * ld + col0;      // (never dereferenced for i < 0)
      // a wavefront's segment is inside the triangle (plain loads), outside it (nothing to read) or crosses the diagonal (one segment
      // per row: per-slot predicates): a scalar branch, so the common case carries no exec-mask juggling
      if (seg0 + 64 * NC - 1 <= i) {
#pragma unroll
        for (int p = 0; p < NP; ++p) w[q][p] = *reinterpret_cast<const double2*>(row + 128 * p);
      } else if (seg0 > i) {
#pragma unroll
        for (int p = 0; p < NP; ++p) w[q][p] = make_double2(0.0, 0.0);
      } else {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          const int col = col0 + 128 * p;        // the pair is inside the triangle iff col <= i, its second half iff col < i
          const double2 v = col <= i ? *reinterpret_cast<const double2*>(row + 128 * p) : make_double2(0.0, 0.0);
          w[q][p] = make_double2(v.x, col < i ? v.y : 0.0);
        }
      }
    }
  };
  auto process = [&](double2 (&w)[RB][NP], const int r0, const int it) __attribute__((always_inline)) {
    // lane q of the first wavefront exchanges row q: its pivot is in flight across the barrier
    const int rq = r0 + (tid < RB ? tid : 0);
    const int iq = rq < r_end ? K - 1 - rq : -1;
    const double dv = (tid < RB && iq >= 0) ? dinv[iq] : 0.0;
#pragma unroll
    for (int q = 0; q < RB; ++q) {
      double s = 0.0;
#pragma unroll
      for (int p = 0; p < NP; ++p) { s += w[q][p].x * ts_zs[(2 * p) * 1024 + tid]; s += w[q][p].y * ts_zs[(2 * p + 1) * 1024 + tid]; }
      s = wave_sum(s);
      if (lane == 0) red[it & 1][q][wave] = s;
    }
    __syncthreads();
    if (tid < RB) {
      const double* rr = red[it & 1][tid];
      const double own = (((rr[0] + rr[1]) + (rr[2] + rr[3])) + ((rr[4] + rr[5]) + (rr[6] + rr[7]))) +
                         (((rr[8] + rr[9]) + (rr[10] + rr[11])) + ((rr[12] + rr[13]) + (rr[14] + rr[15])));
      double u = 0.0;
      if (iq >= 0) {
        unsigned long long* slot = part + (size_t)iq * Q;
        // own != own: a NaN (whatever its payload, the sentinel's included) leaves as the canonical one -- never as "not there yet"
        const unsigned long long own_bits = own != own ? TS_CANON_NAN : (unsigned long long)__double_as_longlong(own);
        __hip_atomic_store(slot + member, own_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the others' parts: all loads of a round in flight together; a round repeats only for those not there yet
        unsigned long long b[Q];
#pragma unroll
        for (int m = 0; m < Q; ++m) b[m] = m == member ? 0ull : __hip_atomic_load(slot + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long long spins = 0;
        for (;;) {
          bool missing = false;
#pragma unroll
          for (int m = 0; m < Q; ++m) missing = missing || (m != member && b[m] == TS_SENTINEL);
          if (!missing) break;
          // still the sentinel after ~seconds, or somebody else gave up (polled every 256th spin): NaN and a raised counter
          // (TailSolve::fail_count) instead of a hang
          ++spins;
          if (spins > (1ll << 22) || (fail && (spins & 255) == 0 && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            if (fail) atomicAdd(fail, 1);
#pragma unroll
            for (int m = 0; m < Q; ++m) if (m != member && b[m] == TS_SENTINEL) b[m] = TS_CANON_NAN;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
#pragma unroll
          for (int m = 0; m < Q; ++m)
            if (m != member && b[m] == TS_SENTINEL) b[m] = __hip_atomic_load(slot + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        double v[Q];
#pragma unroll
        for (int m = 0; m < Q; ++m) v[m] = m == member ? own : __longlong_as_double((long long)b[m]);
        u = Q == 8 ? ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4 % Q] + v[5 % Q]) + (v[6 % Q] + v[7 % Q])) : Q == 4 ? (v[0] + v[1]) + (v[2 % Q] + v[3 % Q]) : v[0] + v[1];
      }
      ush[it & 1][tid] = u * dv;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < RB; ++q) {
      const double vq = ush[it & 1][q];
#pragma unroll
      for (int p = 0; p < NP; ++p) { xa[p].x += vq * w[q][p].x; xa[p].y += vq * w[q][p].y; }
    }
    };
  if constexpr (!PF) {
    double2 w[RB][NP];
    for (int it = 0; it < count; ++it) { load_group(w, row_of(it)); process(w, row_of(it), it); }
  } else {
    // PF: the NEXT group's rows travel while this group's parts go round the members -- the ~2 us exchange no longer stands between two loads
    // (two register buffers: half the rows per exchange of the variant without, the same registers)
    double2 wA[RB][NP], wB[RB][NP];
    if (count > 0) load_group(wA, row_of(0));
    for (int it = 0; it < count; it += 2) {
      if (it + 1 < count) load_group(wB, row_of(it + 1));
      process(wA, row_of(it), it);
      if (it + 1 >= count) break;
      if (it + 2 < count) load_group(wA, row_of(it + 2));
      process(wB, row_of(it + 1), it + 1);
    }
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) { const int col = col0 + 128 * p; if (col < K) *reinterpret_cast<double2*>(P + (size_t)group * K + col) = xa[p]; }
}

// `part` (with Q slots per row): the exchange slots of ts_onepass_group_kernel, reset to the sentinel for the next solve.
// 32 columns x 8 slices per workgroup: slice q sums the partial vectors g = q G/8 ... (q + 1) G/8 - 1 in order (eight loads in flight), the
// slices are added in order through LDS -- a fixed association, four dependent round trips instead of the 32 of one thread per column
// (G = 256 workgroups: 14 us -> 5 us per solve; pendulum / PlanarHand_N=1 / PushBox: 2 - 4 % of an iteration).
__global__ __launch_bounds__(256) void ts_onepass_reduce_kernel(const double* __restrict__ P, int K, int G, double* __restrict__ x,
                                                                unsigned long long* __restrict__ part = nullptr, int Q = 0,
                                                                const int* __restrict__ perm = nullptr) {
  __shared__ double red[8][32];
  const int c = (int)threadIdx.x & 31, q = (int)threadIdx.x >> 5;
  const int col = (int)blockIdx.x * 32 + c;
  const int per = (G + 7) / 8, g0 = q * per, g1 = g0 + per < G ? g0 + per : G;
  double s = 0.0;
  if (col < K) {
    int g = g0;
    for (; g + 8 <= g1; g += 8) {
      double p[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] = P[(size_t)(g + u) * K + col];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += p[u];
    }
    for (; g < g1; ++g) s += P[(size_t)g * K + col];
  }
  red[q][c] = s;
  __syncthreads();
  if (q == 0 && col < K) {
    double t = red[0][c];
#pragma unroll
    for (int u = 1; u < 8; ++u) t += red[u][c];
    x[perm ? perm[col] : col] = t;            // back from the factor's pivoting order
    for (int m = 0; m < Q; ++m) part[(size_t)col * Q + m] = TS_SENTINEL;
  }
}

// ------------------------------------------------------------------------------------------
// Dense LDL^T (no pivoting, as the host factor) of the Schur complement, right-looking, 64-wide block columns:
//   diag  : S_bb = L_bb D_b L_bb^T                           one workgroup
//   panel : Y_ib = S_ib L_bb^-T,  L_ib = Y_ib D_b^-1         one workgroup per 64 rows
//   update: S_ij -= Y_ib L_jb^T  (i >= j > b)                one workgroup per 64 x 64 tile, fp64 MFMA
// Only the lower triangle is referenced.  Every kernel is deterministic.
// ------------------------------------------------------------------------------------------
__global__ void ts_scatter_csr_kernel(const long long* __restrict__ rp, const int* __restrict__ ci, const double* __restrict__ v,
                                      int k, double* __restrict__ S, long long ld) {
  const int row = (int)blockIdx.x;
  if (row >= k) {   // padding rows: identity
    if (threadIdx.x == 0) S[(size_t)row * ld + row] = 1.0;
    return;
  }
  for (long long p = rp[row] + threadIdx.x; p < rp[row + 1]; p += blockDim.x) S[(size_t)row * ld + ci[p]] = v[p];
}

__global__ __launch_bounds__(64) void ts_ldlt_diag_kernel(double* __restrict__ S, long long ld, int b0, double* __restrict__ dvec,
                                                          int* __restrict__ flag) {
  __shared__ double A[64 * 65];
  __shared__ double lcol[64];
  const int r = (int)threadIdx.x;
  const size_t base = (size_t)b0 * ld + b0;
  for (int i = 0; i < 64; ++i) A[i * 65 + r] = S[base + (size_t)i * ld + r];
  for (int j = 0; j < 64; ++j) {
    __syncthreads();
    const double d = A[j * 65 + j];
    double l = 0.0;
    if (r == j) {
      dvec[b0 + j] = d;
      if (d == 0.0 || !(fabs(d) <= 1.7976931348623157e308)) atomicAdd(flag, 1);
    }
    if (r > j) { l = A[r * 65 + j] / d; lcol[r] = l; }
    __syncthreads();
    if (r > j) {
      const double ld_ = l * d;
      for (int c = j + 1; c <= r; ++c) A[r * 65 + c] -= ld_ * lcol[c];
      A[r * 65 + j] = l;
    }
  }
  __syncthreads();
  for (int i = 0; i < 64; ++i) S[base + (size_t)i * ld + r] = A[i * 65 + r];
}

// rows of block-row (b+1+blockIdx.x): y = x L_bb^-T by forward substitution over the 64 columns, l = y / d
__global__ __launch_bounds__(64) void ts_ldlt_panel_kernel(double* __restrict__ S, long long ld, int b0, const double* __restrict__ dvec,
                                                           double* __restrict__ Yp) {
  __shared__ double Lb[64 * 65];
  __shared__ double T[64 * 65];
  const int r = (int)threadIdx.x;
  const int row0 = b0 + 64 + (int)blockIdx.x * 64;
  for (int i = 0; i < 64; ++i) {
    Lb[i * 65 + r] = S[(size_t)(b0 + i) * ld + b0 + r];
    T[i * 65 + r] = S[(size_t)(row0 + i) * ld + b0 + r];
  }
  __syncthreads();
  double y[64];
#pragma unroll
  for (int c = 0; c < 64; ++c) {
    double s = T[r * 65 + c];
#pragma unroll
    for (int j = 0; j < c; ++j) s -= y[j] * Lb[c * 65 + j];
    y[c] = s;
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 64; ++c) { T[r * 65 + c] = y[c]; }
  __syncthreads();
  for (int i = 0; i < 64; ++i) {
    const double yv = T[i * 65 + r];
    Yp[(size_t)(row0 + i) * 64 + r] = yv;
    S[(size_t)(row0 + i) * ld + b0 + r] = yv / dvec[b0 + r];
  }
}

// trailing update, tile (ti >= tj) of the block rows / columns after b: C -= Y_ti * L_tj^T  (k = 64)
__global__ __launch_bounds__(256) void ts_ldlt_update_kernel(double* __restrict__ S, long long ld, int b0, const double* __restrict__ Yp) {
  constexpr int LDS = TS_TM + 16, WT = TS_TM / 2, NTW = WT / 16;
  __shared__ double As[64 * LDS];   // As[k][row]
  __shared__ double Bs[64 * LDS];   // Bs[k][col]
  int ti, tj;
  {
    const int e = (int)blockIdx.x;
    int t = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
    while (t * (t + 1) / 2 > e) --t;
    while ((t + 1) * (t + 2) / 2 <= e) ++t;
    ti = t; tj = e - t * (t + 1) / 2;
  }
  const int row0 = b0 + 64 + ti * 64, col0 = b0 + 64 + tj * 64;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wy = wave >> 1, wx = wave & 1, r16 = lane & 15, kk = lane >> 4;
  // both operands are 64 x 64 row-major blocks [x][k], staged transposed: thread t: x = t/4, k = (t%4)*16 .. +15
  {
    const int x = tid >> 2, kq = (tid & 3) * 16;
    const double* ya = Yp + (size_t)(row0 + x) * 64 + kq;
    const double* lb = S + (size_t)(col0 + x) * ld + b0 + kq;
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
      const double2 va = *reinterpret_cast<const double2*>(ya + c);
      const double2 vb = *reinterpret_cast<const double2*>(lb + c);
      As[(kq + c) * LDS + x] = va.x; As[(kq + c + 1) * LDS + x] = va.y;
      Bs[(kq + c) * LDS + x] = vb.x; Bs[(kq + c + 1) * LDS + x] = vb.y;
    }
  }
  __syncthreads();
  ts_v4f64 acc[NTW][NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = ts_v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int ks = 0; ks < 64; ks += 4) {
    double af[NTW], bf[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      af[t] = As[(ks + kk) * LDS + wy * WT + t * 16 + r16];
      bf[t] = Bs[(ks + kk) * LDS + wx * WT + t * 16 + r16];
    }
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + wy * WT + i * 16 + kk + 4 * r, col = col0 + wx * WT + j * 16 + r16;
        S[(size_t)row * ld + col] -= acc[i][j][r];
      }
}

// ------------------------------------------------------------------------------------------
// The same factorisation WITH DIAGONAL PIVOTING (round 6; option tail_pivot, default on): P S P^T = L D L^T with the largest remaining
// diagonal entry eliminated next (LAPACK dpstrf's rule; for a positive semidefinite S every |L_ij| <= 1).  Why: the tail is applied as
// an EXPLICIT inverse W = inv(L), accurate to u cond(L); the Schur complement of a large moment relaxation is nearly singular, the
// unpivoted factor then carries columns of size 1 / sqrt(pivot), and the lost digits surfaced as 1e-8 ... 1e-7 in the primal objective
// (NOTEBOOK.md "Round 6"; one refinement step per triangular solve -- option tail_refine -- removed them at 6 x the bytes per solve).
// With pivoting the small pivots come LAST and L stays bounded: the explicit inverse is accurate at no cost per solve.
// Per column two launches: (A) one workgroup finds the pivot among rows [j, k) of the running diagonal `dg`, swaps it into place --
// symmetric swap on the lower triangle, permutation and diagonal with it -- and records the pivot; (B) every row below forms its entry
// of the column from the block's deferred updates (dot over <= 63 earlier columns of the block) and updates its diagonal.  Per block of
// 64 columns the trailing matrix takes the usual rank-64 update (ts_ldlt_update_kernel) and the diagonal is re-read from it.
// Padding rows (identity, >= k) never move.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void ts_piv_select_kernel(int k, int j, const double* __restrict__ dg, int* __restrict__ piv) {
  __shared__ double bv[16];
  __shared__ int bi[16];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // argmax of dg over [j, k) (ties: the smallest index, so the choice does not depend on the thread layout)
  double best = -1.7976931348623157e308;
  int at = j;
  for (int i = j + tid; i < k; i += 1024) { const double v = dg[i]; if (v > best) { best = v; at = i; } }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const double ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(at, o, 64);
    if (ov > best || (ov == best && oi < at)) { best = ov; at = oi; }
  }
  if (lane == 0) { bv[wave] = best; bi[wave] = at; }
  __syncthreads();
  if (tid == 0) {
    double b = bv[0]; int a = bi[0];
    for (int w = 1; w < 16; ++w) if (bv[w] > b || (bv[w] == b && bi[w] < a)) { b = bv[w]; a = bi[w]; }
    piv[0] = j < k ? a : j;
  }
}
// symmetric swap of indices j < p = piv[0] in the lower triangle, one thread per index i: row segments left of j, the bent segment between
// j and p, column segments below p; the diagonal, the running diagonal and the permutation with them
__global__ __launch_bounds__(256) void ts_piv_swap_kernel(double* __restrict__ S, long long ld, int K, int j, const int* __restrict__ piv,
                                                          double* __restrict__ dg, int* __restrict__ perm) {
  const int p = piv[0];
  if (p == j) return;
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= K) return;
  double *a, *b;
  if (i < j) { a = S + (size_t)j * ld + i; b = S + (size_t)p * ld + i; }
  else if (i == j) {
    a = S + (size_t)j * ld + j; b = S + (size_t)p * ld + p;
    const double g = dg[j]; dg[j] = dg[p]; dg[p] = g;
    const int q = perm[j]; perm[j] = perm[p]; perm[p] = q;
  }
  else if (i < p) { a = S + (size_t)i * ld + j; b = S + (size_t)p * ld + i; }
  else if (i == p) return;                                    // S[p][j] keeps its place
  else { a = S + (size_t)i * ld + j; b = S + (size_t)i * ld + p; }
  const double t = *a; *a = *b; *b = t;
}
// column j of L below the diagonal from the block's deferred updates: l_i = (S_ij - sum_{c = b0}^{j-1} L_ic d_c L_jc) / d_j
__global__ __launch_bounds__(256) void ts_piv_column_kernel(double* __restrict__ S, long long ld, int K, int k, int j, int b0, double* __restrict__ dg,
                                                            double* __restrict__ dvec, int* __restrict__ flag) {
  __shared__ double yj[64];
  const int nc = j - b0;
  if ((int)threadIdx.x < nc) yj[threadIdx.x] = S[(size_t)j * ld + b0 + threadIdx.x] * dvec[b0 + threadIdx.x];
  __syncthreads();
  const double d = dg[j];                                     // the pivot: nobody writes dg[j] any more
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    dvec[j] = d;
    if (d == 0.0 || !(fabs(d) <= 1.7976931348623157e308)) atomicAdd(flag, 1);
  }
  const int i = j + 1 + (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= K) return;
  const double* row = S + (size_t)i * ld + b0;
  double s = row[nc];
  for (int c = 0; c < nc; ++c) s -= row[c] * yj[c];
  const double l = i < k ? s / d : 0.0;            // (padding rows are decoupled: exact zeros)
  S[(size_t)i * ld + j] = l;
  if (i < k) dg[i] -= l * l * d;
}
// Y = L_21 D of the finished block for the trailing update (formed at the END of the block: rows still swap while its columns are eliminated)
__global__ void ts_piv_yp_kernel(const double* __restrict__ S, long long ld, int K, int b0, const double* __restrict__ dvec, double* __restrict__ Yp) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int i = b0 + 64 + (int)(t >> 6), c = (int)(t & 63);
  if (i < K) Yp[(size_t)i * 64 + c] = S[(size_t)i * ld + b0 + c] * dvec[b0 + c];
}
__global__ void ts_piv_diag_kernel(const double* __restrict__ S, long long ld, int K, int from, double* __restrict__ dg, int* __restrict__ perm, int init) {
  const int i = from + (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i < K) { dg[i] = S[(size_t)i * ld + i]; if (init) perm[i] = i; }
}
// the unit diagonal of the finished factor's block columns is implicit for ts_unit_lower_inverse; dd: pivots, perm: position -> original index
int ts_ldlt_factor_pivoted(double* dS, int K, int k, double* dd, double* Yp, double* dg, int* perm, int* dflag, hipStream_t st) {
  const long long ld = K;
  const int nbk = K / 64;
  int* piv_d = nullptr;                                         // the pivot's index travels from the search to the swap on the device
  CUADMM_HIP_TRY(hipMalloc(&piv_d, sizeof(int)));
  int* const piv = piv_d;
  hipLaunchKernelGGL(ts_piv_diag_kernel, dim3((K + 255) / 256), dim3(256), 0, st, dS, ld, K, 0, dg, perm, 1);
  for (int b = 0; b < nbk; ++b) {
    const int b0 = b * 64, T = nbk - b - 1;
    for (int j = b0; j < b0 + 64; ++j) {
      if (j < k) {                                               // (padding columns keep their place)
        hipLaunchKernelGGL(ts_piv_select_kernel, dim3(1), dim3(1024), 0, st, k, j, dg, piv);
        hipLaunchKernelGGL(ts_piv_swap_kernel, dim3((K + 255) / 256), dim3(256), 0, st, dS, ld, K, j, piv, dg, perm);
      }
      // (the last column has no rows below it: the launch still records its pivot)
      hipLaunchKernelGGL(ts_piv_column_kernel, dim3((unsigned)std::max(1, (K - j - 1 + 255) / 256)), dim3(256), 0, st, dS, ld, K, k, j, b0, dg, dd, dflag);
    }
    if (T > 0) {
      hipLaunchKernelGGL(ts_piv_yp_kernel, dim3((unsigned)(((long long)(K - b0 - 64) * 64 + 255) / 256)), dim3(256), 0, st, dS, ld, K, b0, dd, Yp);
      hipLaunchKernelGGL(ts_ldlt_update_kernel, dim3(T * (T + 1) / 2), dim3(256), 0, st, dS, ld, b0, Yp);
      hipLaunchKernelGGL(ts_piv_diag_kernel, dim3((K - b0 - 64 + 255) / 256), dim3(256), 0, st, dS, ld, K, b0 + 64, dg, perm, 0);
    }
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  { hipError_t e2 = hipFree(piv_d); (void)e2; }
  if (e != hipSuccess) { set_error("tail_solve: %s", hipGetErrorString(e)); return CUADMM_ERR_NO_DEVICE; }
  return CUADMM_OK;
}

// pinv_tol > 0 (option tail_pinv_tol, an experiment: see tail_solve.h): a pivot below it in magnitude counts as zero
__global__ void ts_dinv_kernel(const double* __restrict__ d, double* __restrict__ dinv, int K, double pinv_tol) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i < K) dinv[i] = fabs(d[i]) < pinv_tol ? 0.0 : 1.0 / d[i];
}

int ts_gemm(int M, int N, int Kd, double alpha, const double* A, long long lda, long long sA, const double* B, long long ldb,
                   long long sB, double* C, long long ldc, long long sC, int batch, hipStream_t st) {
  if (batch <= 0 || M <= 0 || N <= 0) return CUADMM_OK;
  hipLaunchKernelGGL(ts_gemm_nn_kernel, dim3((M / TS_TM) * (N / TS_TM), batch), dim3(256), 0, st, M, N, Kd, alpha, A, lda, sA, B, ldb, sB,
                     C, ldc, sC);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

void TailSolve::release() {
  if (compact) { W = Wc; Wc = nullptr; compact = false; ldc = 0; i_lo = 0; i_hi = -1; }     // (W was an offset view of Wc)
  for (void* p : {(void*)W, (void*)Wt, (void*)dinv, (void*)vin, (void*)vmid, (void*)xpart, (void*)part, (void*)Lm, (void*)Lt, (void*)t1, (void*)t2, (void*)perm_d, (void*)pinv_d, (void*)pv1, (void*)pv2}) if (p) { hipError_t e = hipFree(p); (void)e; }
  part = nullptr; Lm = Lt = t1 = t2 = nullptr; perm_d = nullptr; pinv_d = nullptr; pv1 = pv2 = nullptr; vin_pivot = false;
  if (d_fail) { hipError_t e = hipFree(d_fail); (void)e; d_fail = nullptr; }
  if (h_vec) { hipError_t e = hipHostFree(h_vec); (void)e; }
  W = Wt = dinv = vin = vmid = h_vec = xpart = nullptr;
  attr_set = false;
  group_retired = false;
  k = K = 0;
  resident_bytes = 0;
  inv_resid = -1.0;
}

int TailSolve::alloc(int k_) {
  release();
  k = k_;
  K = tail_padded(k);
  static_assert(TS_TM == 64, "tail_padded (tail_solve.h) pads to the tile of the dense kernels");
  const size_t sz = (size_t)K * K;
  CUADMM_HIP_TRY(hipMalloc(&W, sizeof(double) * sz));
  CUADMM_HIP_TRY(hipMalloc(&Wt, sizeof(double) * sz));
  CUADMM_HIP_TRY(hipMalloc(&dinv, sizeof(double) * (size_t)K));
  CUADMM_HIP_TRY(hipMalloc(&vin, sizeof(double) * (size_t)K));
  CUADMM_HIP_TRY(hipMalloc(&vmid, sizeof(double) * (size_t)K));
  {
    int dev = 0;
    hipDeviceProp_t prop;
    CUADMM_HIP_TRY(hipGetDevice(&dev));
    CUADMM_HIP_TRY(hipGetDeviceProperties(&prop, dev));
    n_wg = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  CUADMM_HIP_TRY(hipMalloc(&xpart, sizeof(double) * (size_t)K * (size_t)n_wg));
  if (K > 18432 && K <= 65536) {        // exchange slots of the row-sharing one-pass kernel (four workgroups per row up to 32 768 columns, eight beyond)
    CUADMM_HIP_TRY(hipMalloc(&part, sizeof(unsigned long long) * (size_t)K * 8));
    CUADMM_HIP_TRY(hipMalloc(&d_fail, sizeof(int)));
    CUADMM_HIP_TRY(hipMemset(d_fail, 0, sizeof(int)));
    hipLaunchKernelGGL(ts_fill_u64_kernel, dim3((unsigned)(((size_t)K * 8 + 255) / 256)), dim3(256), 0, nullptr, part, (size_t)K * 8, TS_SENTINEL);
    CUADMM_HIP_TRY(hipGetLastError());
    CUADMM_HIP_TRY(hipDeviceSynchronize());
  }
  CUADMM_HIP_TRY(hipHostMalloc(&h_vec, sizeof(double) * (size_t)K, hipHostMallocDefault));
  resident_bytes = 8.0 * (2.0 * (double)sz + 3.0 * K + (double)K * n_wg) + (part ? 64.0 * K : 0.0);
  return CUADMM_OK;
}

// W = inv(dL) by recursive doubling (dL: K x K unit lower triangular on the device, row-major, strict upper part never read;
// dT: K x K scratch of the products).  Asynchronous on `st`.
int ts_unit_lower_inverse(const double* dL, double* W, double* dT, int K, hipStream_t st) {
  const long long ld = K;
  CUADMM_HIP_TRY(hipMemsetAsync(W, 0, sizeof(double) * (size_t)K * K, st));
  hipLaunchKernelGGL(ts_diag_inverse_kernel, dim3(K / 64), dim3(64), 0, st, dL, W, ld);
  CUADMM_HIP_TRY(hipGetLastError());
  int rc = CUADMM_OK;
  for (long long h = 64; h < K && !rc; h *= 2) {
    // groups [a, a+2h): W21 = -W22 * (L21 * W11); the second half may be short (h2 < h) in the last group only
    const int full = (int)(K / (2 * h));
    const long long gs = 2 * h * (ld + 1);   // element stride between groups (diagonal step)
    if (full > 0) {
      rc = ts_gemm((int)h, (int)h, (int)h, 1.0, dL + h * ld, ld, gs, W, ld, gs, dT + h * ld, ld, gs, full, st);
      if (!rc) rc = ts_gemm((int)h, (int)h, (int)h, -1.0, W + h * (ld + 1), ld, gs, dT + h * ld, ld, gs, W + h * ld, ld, gs, full, st);
    }
    const long long a = (long long)full * 2 * h;
    const long long h2 = K - a - h;
    if (!rc && h2 > 0) {
      const size_t o = (size_t)a * (ld + 1);
      rc = ts_gemm((int)h2, (int)h, (int)h, 1.0, dL + o + h * ld, ld, 0, W + o, ld, 0, dT + o + h * ld, ld, 0, 1, st);
      if (!rc) rc = ts_gemm((int)h2, (int)h, (int)h2, -1.0, W + o + h * (ld + 1), ld, 0, dT + o + h * ld, ld, 0, W + o + h * ld, ld, 0, 1, st);
    }
  }
  return rc;
}

int ts_transpose(const double* src, double* dst, int K, hipStream_t st) {
  hipLaunchKernelGGL(ts_transpose_kernel, dim3(K / 32, K / 32), dim3(256), 0, st, src, dst, (long long)K);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

// in-place dense LDL^T of a K x K symmetric matrix (row-major, lower triangle referenced and overwritten by the unit factor; dd: the
// pivots; Yp: K x 64 scratch; dflag: counts zero / non-finite pivots).  Asynchronous on `st`.
int ts_ldlt_factor(double* dS, int K, double* dd, double* Yp, int* dflag, hipStream_t st) {
  const long long ld = K;
  const int nbk = K / 64;
  for (int b = 0; b < nbk; ++b) {
    const int b0 = b * 64, T = nbk - b - 1;
    hipLaunchKernelGGL(ts_ldlt_diag_kernel, dim3(1), dim3(64), 0, st, dS, ld, b0, dd, dflag);
    if (T > 0) {
      hipLaunchKernelGGL(ts_ldlt_panel_kernel, dim3(T), dim3(64), 0, st, dS, ld, b0, dd, Yp);
      hipLaunchKernelGGL(ts_ldlt_update_kernel, dim3(T * (T + 1) / 2), dim3(256), 0, st, dS, ld, b0, Yp);
    }
  }
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

// W = inv(dL), then Wt = W^T (Wt doubles as the scratch matrix of the products)
int TailSolve::invert(const double* dL, hipStream_t st) {
  int rc = ts_unit_lower_inverse(dL, W, Wt, K, st);
  if (rc) return rc;
  if ((rc = ts_transpose(W, Wt, K, st))) return rc;
  CUADMM_HIP_TRY(hipStreamSynchronize(st));
  return CUADMM_OK;
}

// L22: k x k dense row-major unit lower triangular (host), D2: k pivots (host)
int TailSolve::build(const double* L22, const double* D2, int k_, hipStream_t st) {
  const auto t0 = std::chrono::steady_clock::now();
  int rc = alloc(k_);
  if (rc) return rc;
  const long long ld = K;
  const size_t sz = (size_t)K * K;
  double* dL = nullptr;
  CUADMM_HIP_TRY(hipMalloc(&dL, sizeof(double) * sz));
  // upload: rows of L22 (leading dimension k) into the padded matrix (identity in the padding)
  CUADMM_HIP_TRY(hipMemsetAsync(dL, 0, sizeof(double) * sz, st));
  if ((rc = staged_h2d_2d(dL, sizeof(double) * (size_t)ld, L22, sizeof(double) * (size_t)k, sizeof(double) * (size_t)k, (size_t)k, st))) {
    hipError_t e = hipFree(dL); (void)e;
    return rc;
  }
  {
    std::vector<double> di((size_t)K, 1.0);
    for (int i = 0; i < k; ++i) di[i] = 1.0 / D2[i];
    if ((rc = staged_h2d(dinv, di.data(), sizeof(double) * (size_t)K, st))) { hipError_t e = hipFree(dL); (void)e; return rc; }
  }
  rc = invert(dL, st);   // the padding rows of dL are zero: the diagonal is implicit
  { hipError_t e = hipFree(dL); (void)e; }
  build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (rc) release();
  return rc;
}

// Schur complement (lower triangle with diagonal, CSR over the k tail rows, host pointers) -> dense LDL^T on the GPU
// -> W = inv(L22), dinv = 1 / D2
int TailSolve::build_from_schur(const long long* row_ptr, const int* col, const double* val, int k_, hipStream_t st) {
  const auto t0 = std::chrono::steady_clock::now();
  int rc = alloc(k_);
  if (rc) return rc;
  const long long ld = K;
  const size_t sz = (size_t)K * K;
  const long long nnz = row_ptr[k];
  double *dS = nullptr, *dval = nullptr, *dd = nullptr, *Yp = nullptr;
  long long* drp = nullptr;
  int *dci = nullptr, *dflag = nullptr;
  auto cleanup = [&]() {
    for (void* p : {(void*)dS, (void*)dval, (void*)dd, (void*)Yp, (void*)drp, (void*)dci, (void*)dflag}) if (p) { hipError_t e = hipFree(p); (void)e; }
  };
  hipError_t e = hipSuccess;
  if (e == hipSuccess) e = hipMalloc(&dS, sizeof(double) * sz);
  if (e == hipSuccess) e = hipMalloc(&dval, sizeof(double) * (size_t)std::max<long long>(nnz, 1));
  if (e == hipSuccess) e = hipMalloc(&dci, sizeof(int) * (size_t)std::max<long long>(nnz, 1));
  if (e == hipSuccess) e = hipMalloc(&drp, sizeof(long long) * ((size_t)k + 1));
  if (e == hipSuccess) e = hipMalloc(&dd, sizeof(double) * (size_t)K);
  if (e == hipSuccess) e = hipMalloc(&Yp, sizeof(double) * (size_t)K * 64);
  if (e == hipSuccess) e = hipMalloc(&dflag, sizeof(int));
  if (e == hipSuccess) e = hipMemsetAsync(dS, 0, sizeof(double) * sz, st);
  if (e == hipSuccess) e = hipMemsetAsync(dflag, 0, sizeof(int), st);
  if (e == hipSuccess && (staged_h2d(drp, row_ptr, sizeof(long long) * ((size_t)k + 1), st) ||
                          (nnz > 0 && (staged_h2d(dci, col, sizeof(int) * (size_t)nnz, st) || staged_h2d(dval, val, sizeof(double) * (size_t)nnz, st))))) {
    cleanup(); release(); return CUADMM_ERR_NO_DEVICE;
  }
  if (e != hipSuccess) { set_error("tail_solve: %s", hipGetErrorString(e)); cleanup(); release(); return e == hipErrorOutOfMemory ? CUADMM_ERR_INVALID : CUADMM_ERR_NO_DEVICE; }
  hipLaunchKernelGGL(ts_scatter_csr_kernel, dim3(K), dim3(256), 0, st, drp, dci, dval, k, dS, ld);
  if (pivot) {
    double* dg = nullptr;
    if (hipMalloc(&perm_d, sizeof(int) * (size_t)K) != hipSuccess || hipMalloc(&pv1, sizeof(double) * (size_t)K) != hipSuccess ||
        hipMalloc(&pv2, sizeof(double) * (size_t)K) != hipSuccess || hipMalloc(&dg, sizeof(double) * (size_t)K) != hipSuccess) {
      if (dg) { hipError_t e2 = hipFree(dg); (void)e2; }
      set_error("tail_solve: out of device memory"); cleanup(); release(); return CUADMM_ERR_INVALID;
    }
    rc = ts_ldlt_factor_pivoted(dS, K, k, dd, Yp, dg, perm_d, dflag, st);
    if (!rc && hipMalloc(&pinv_d, sizeof(int) * (size_t)K) == hipSuccess)      // (optional: without it the kernels gather z as before)
      hipLaunchKernelGGL(ts_perm_inverse_kernel, dim3((K + 255) / 256), dim3(256), 0, st, perm_d, pinv_d, K);
    if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = CUADMM_ERR_NO_DEVICE;
    { hipError_t e2 = hipFree(dg); (void)e2; }
    if (rc) { cleanup(); release(); return rc; }
  } else if ((rc = ts_ldlt_factor(dS, K, dd, Yp, dflag, st))) { cleanup(); release(); return rc; }
  hipLaunchKernelGGL(ts_dinv_kernel, dim3((K + 255) / 256), dim3(256), 0, st, dd, dinv, K, pinv_tol);
  int hflag = 0;
  e = hipGetLastError();
  if (e == hipSuccess && staged_d2h(&hflag, dflag, sizeof(int), st)) e = hipErrorUnknown;      // staged: never a runtime copy into pageable memory
  factor_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (e != hipSuccess) { set_error("tail_solve: %s", hipGetErrorString(e)); cleanup(); release(); return CUADMM_ERR_NO_DEVICE; }
  if (hflag) { set_error("Factorization fails! (%d zero or non-finite pivots in the dense tail of A*A^T)", hflag); cleanup(); release(); return CUADMM_ERR_FACTOR; }
  rc = invert(dS, st);
  if (!rc) {
    // How good is the explicit inverse?  rho = || z - L (W z) ||_inf / || z ||_inf for a fixed pseudo-random z: ~ u cond(L).  A Schur complement
    // that is nearly singular gives L columns of size 1 / sqrt(pivot) (PushBox_N=50 at an 8 448-column tail: rho = see profiles/r06_tail_refine.log)
    std::vector<double> zh((size_t)K, 0.0);
    unsigned long long seed = 0x243f6a8885a308d3ull;
    for (int i = 0; i < k; ++i) { seed = seed * 6364136223846793005ull + 1442695040888963407ull; zh[i] = 0.5 + (double)(seed >> 11) * (1.0 / 9007199254740992.0); }
    double* tmp = nullptr;
    if (hipMalloc(&tmp, sizeof(double) * (size_t)K) == hipSuccess && staged_h2d(vin, zh.data(), sizeof(double) * (size_t)K, st) == CUADMM_OK) {
      hipLaunchKernelGGL(ts_tri_gemv_kernel<true>, dim3((K + 3) / 4), dim3(256), 0, st, W, (long long)K, K, vin, (const double*)nullptr, vmid, 0, 1 << 30);
      hipLaunchKernelGGL(ts_tri_resid_kernel<true>, dim3((K + 3) / 4), dim3(256), 0, st, dS, (long long)K, K, vin, vmid, tmp);
      std::vector<double> rh((size_t)K);
      if (hipGetLastError() == hipSuccess && staged_d2h(rh.data(), tmp, sizeof(double) * (size_t)K, st) == CUADMM_OK) {
        double m = 0.0;
        for (int i = 0; i < k; ++i) m = std::max(m, std::fabs(rh[i]));
        inv_resid = m;                       // || z ||_inf ~ 1.5
      }
    }
    if (tmp) { hipError_t e2 = hipFree(tmp); (void)e2; }
  }
  if (!rc && refine) {   // the factor itself stays: L (strictly lower part of dS) and its transpose
    Lm = dS; dS = nullptr;
    if (hipMalloc(&Lt, sizeof(double) * sz) != hipSuccess || hipMalloc(&t1, sizeof(double) * (size_t)K) != hipSuccess || hipMalloc(&t2, sizeof(double) * (size_t)K) != hipSuccess) rc = CUADMM_ERR_INVALID;
    if (!rc) rc = ts_transpose(Lm, Lt, K, st);
    if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = CUADMM_ERR_NO_DEVICE;
    if (!rc) resident_bytes += 16.0 * (double)sz + 16.0 * K;
  }
  cleanup();
  build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (rc) release();
  return rc;
}

// workgroups of the four-per-row kernel that gave up waiting for a partner (0 unless the device could not keep a group resident for
// seconds); synchronises `st`
int TailSolve::fail_count(hipStream_t st) {
  if (!d_fail) return 0;
  int h = 0;
  if (hipStreamSynchronize(st) != hipSuccess || staged_d2h(&h, d_fail, sizeof(int), st)) return -1;
  return h;
}

// Reads the counter and, when it is raised, clears it, resets the exchange slots and retires the four-workgroups-per-row kernel for
// the rest of this object's life (apply() then runs the two triangular GEMVs, which need nothing co-resident): the caller reports
// the lost solve once, the handle stays usable.  Returns the count that was found (-1: the device could not be read).
int TailSolve::take_failure(hipStream_t st) {
  const int h = fail_count(st);
  if (h <= 0) return h;
  group_retired = true;
  if (hipMemsetAsync(d_fail, 0, sizeof(int), st) != hipSuccess) return -1;
  if (part) {
    const size_t n = (size_t)K * 8;
    hipLaunchKernelGGL(ts_fill_u64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, n, TS_SENTINEL);
  }
  if (hipStreamSynchronize(st) != hipSuccess) return -1;
  return h;
}

// x2 = L22^-T D2^-1 L22^-1 z2, host vector in place (k doubles); synchronous on `st`
int TailSolve::solve(double* z2, hipStream_t st) {
  if (!W) { set_error("tail_solve: not built"); return CUADMM_ERR_INVALID; }
  std::copy(z2, z2 + k, h_vec);
  std::fill(h_vec + k, h_vec + K, 0.0);
  CUADMM_HIP_TRY(hipMemcpyAsync(vin, h_vec, sizeof(double) * (size_t)K, hipMemcpyHostToDevice, st));
  { int rc_ = apply(st); if (rc_) return rc_; }
  CUADMM_HIP_TRY(hipMemcpyAsync(h_vec, vin, sizeof(double) * (size_t)k, hipMemcpyDeviceToHost, st));
  CUADMM_HIP_TRY(hipStreamSynchronize(st));
  std::copy(h_vec, h_vec + k, z2);
  return CUADMM_OK;
}

// the same on a right-hand side that is already in `vin` (written by lead_tail_rhs_kernel): nothing crosses PCIe
// does apply() read z through a kernel that can take it in the factor's order (the one-pass kernels)?  Mirrors the branches of apply_rows.
bool TailSolve::linear_z_ok() const {
  if (!perm_d || !pinv_d || !one_pass || !xpart || (refine && Lm)) return false;
  const int nc = (K + 1023) / 1024;
  const size_t lds = sizeof(double) * 1024 * (size_t)(nc + (nc & 1));
  if (lds <= kMaxLdsBytes - 1024 && nc <= 20) return true;
  return part && K <= 65536 && !group_retired;
}
// where entry i of z goes when the producer writes it for the next solve_device (null: in place, the kernels gather).  The producer that
// uses it sets vin_pivot; one solve later the flag is down again.
const int* TailSolve::z_scatter() const { return linear_z_ok() ? pinv_d : nullptr; }

int TailSolve::solve_device(hipStream_t st) {
  if (!W) { set_error("tail_solve: not built"); return CUADMM_ERR_INVALID; }
  return apply(st);
}

// second pass of the two-GEMV fallback on a compact shard: P[chunk][j] = sum over this chunk's rows i >= j of W[i][j] v[i] (thread per column,
// rows in order: deterministic); the chunks are summed by ts_onepass_reduce_kernel
__global__ __launch_bounds__(256) void ts_colacc_kernel(const double* __restrict__ W, long long ld, int K, const double* __restrict__ v, double* __restrict__ P,
                                                        int i_lo, int i_hi, int nchunk) {
  const int j = (int)blockIdx.x * 256 + (int)threadIdx.x, ch = (int)blockIdx.y;
  if (j >= K) return;
  const int rows = i_hi - i_lo + 1, per = (rows + nchunk - 1) / nchunk;
  const int a = i_lo + ch * per, b = a + per - 1 < i_hi ? a + per - 1 : i_hi;
  double s = 0.0;
  for (int i = a > j ? a : j; i <= b; ++i) s += W[(long long)i * ld + j] * v[i];
  P[(size_t)ch * K + j] = s;
}

int TailSolve::keep_shard(int rank, int world, hipStream_t st) {
  if (!W || compact || refine || world <= 1) return CUADMM_OK;
  const int r_b = tail_shard_bound(K, rank, world), r_e = tail_shard_bound(K, rank + 1, world);
  shard_rank = rank; shard_world = world;
  i_lo = K - r_e; i_hi = K - 1 - r_b;                      // matrix rows of the kernels' rows [r_b, r_e)
  const int rows = i_hi - i_lo + 1;
  ldc = rows > 0 ? ((long long)i_hi + 1 + 63) / 64 * 64 : 64;
  CUADMM_HIP_TRY(hipMalloc(&Wc, sizeof(double) * (size_t)std::max(rows, 1) * (size_t)ldc));
  if (rows > 0)
    CUADMM_HIP_TRY(hipMemcpy2DAsync(Wc, sizeof(double) * (size_t)ldc, W + (size_t)i_lo * K, sizeof(double) * (size_t)K, sizeof(double) * (size_t)ldc, (size_t)rows,
                                    hipMemcpyDeviceToDevice, st));
  CUADMM_HIP_TRY(hipStreamSynchronize(st));
  { hipError_t e = hipFree(W); (void)e; e = hipFree(Wt); (void)e; }
  Wt = nullptr;
  W = Wc - (long long)i_lo * ldc;                          // row i of the triangle at W + i * ldc; only rows i_lo .. i_hi exist
  compact = true;
  resident_bytes += 8.0 * (double)std::max(rows, 1) * (double)ldc - 16.0 * (double)K * K;
  return CUADMM_OK;
}

// the two triangular GEMVs with one refinement step each against the factor (experiment, tail_solve.h)
int TailSolve::apply_refined(double* vin, hipStream_t st) {      // (vin: the vector in the factor's order, in place)
  const dim3 gr((K + 3) / 4), bl(256), g1((K + 255) / 256);
  hipLaunchKernelGGL(ts_tri_gemv_kernel<true>, gr, bl, 0, st, W, (long long)K, K, vin, (const double*)nullptr, vmid, 0, 1 << 30);      // u = W z
  hipLaunchKernelGGL(ts_tri_resid_kernel<true>, gr, bl, 0, st, Lm, (long long)K, K, vin, vmid, t1);                                    // r = z - L u
  hipLaunchKernelGGL(ts_tri_gemv_kernel<true>, gr, bl, 0, st, W, (long long)K, K, t1, (const double*)nullptr, t2, 0, 1 << 30);         // W r
  hipLaunchKernelGGL(ts_axpy1_kernel<true>, g1, bl, 0, st, vmid, t2, dinv, K);                                                          // v = D^-1 (u + W r)
  hipLaunchKernelGGL(ts_tri_gemv_kernel<false>, gr, bl, 0, st, Wt, (long long)K, K, vmid, (const double*)nullptr, vin, 0, 1 << 30);    // x = W^T v
  hipLaunchKernelGGL(ts_tri_resid_kernel<false>, gr, bl, 0, st, Lt, (long long)K, K, vmid, vin, t1);                                   // s = v - L^T x
  hipLaunchKernelGGL(ts_tri_gemv_kernel<false>, gr, bl, 0, st, Wt, (long long)K, K, t1, (const double*)nullptr, t2, 0, 1 << 30);       // W^T s
  hipLaunchKernelGGL(ts_axpy1_kernel<false>, g1, bl, 0, st, vin, t2, (const double*)nullptr, K);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

// Which (NC, RB, D) exist: RB NC (D + 1) doubles of row buffers + NC accumulators per thread within ~104 of the 128 registers a 1024-thread
// workgroup leaves a wavefront (the rest: z pairs in flight, addresses, the reduction).  Measured by the compiler's own count, not guessed:
// every instantiation below builds without scratch (tools: -Rpass-analysis=kernel-resource-usage).
template <int NC, int RB, int D, bool ZREG>
constexpr bool ts_onepass_fits = (D <= 1 || (RB == 1 && D <= 3 && NC <= 10)) && NC * RB * (D + 1) + NC + (ZREG ? NC : 0) <= 52;
template <int NC, int RB, int D, class L>
static int ts_onepass_launch_one(int order, bool zreg, L& launch) {
  if constexpr (ts_onepass_fits<NC, RB, D, true>) {
    if (zreg) return order == 2 ? launch(ts_onepass_kernel<NC, RB, D, false, 1024, 2, true>) : order ? launch(ts_onepass_kernel<NC, RB, D, false, 1024, 1, true>) : launch(ts_onepass_kernel<NC, RB, D, false, 1024, 0, true>);
  }
  if constexpr (ts_onepass_fits<NC, RB, D, false>) {
    return order == 2 ? launch(ts_onepass_kernel<NC, RB, D, false, 1024, 2>) : order ? launch(ts_onepass_kernel<NC, RB, D, false, 1024, 1>) : launch(ts_onepass_kernel<NC, RB, D, false, 1024, 0>);
  } else {
    (void)order; (void)launch; (void)zreg;
    return -1;
  }
}
template <int NC, class L>
static int ts_onepass_launch(int rb, int dep, int order, bool zreg, L& launch) {
  if (rb == 2) {
    switch (dep) {
      case 0: return ts_onepass_launch_one<NC, 2, 0>(order, zreg, launch);
      case 1: return ts_onepass_launch_one<NC, 2, 1>(order, zreg, launch);
      default: return ts_onepass_launch_one<NC, 2, 2>(order, zreg, launch);
    }
  }
  switch (dep) {
    case 0: return ts_onepass_launch_one<NC, 1, 0>(order, zreg, launch);
    case 1: return ts_onepass_launch_one<NC, 1, 1>(order, zreg, launch);
    case 2: return ts_onepass_launch_one<NC, 1, 2>(order, zreg, launch);
    default: return ts_onepass_launch_one<NC, 1, 3>(order, zreg, launch);
  }
}

// vin <- W^T diag(dinv) W vin
int TailSolve::apply(hipStream_t st) {
  // (refined: every rank of a sharded engine applies the WHOLE tail -- the result is replicated, no reduction)
  if (refine && Lm) {
    shard_rows = K; shard_bytes = 24.0 * (double)K * K;
    if (!perm_d) return apply_refined(vin, st);
    hipLaunchKernelGGL(ts_perm_kernel<true>, dim3((K + 255) / 256), dim3(256), 0, st, pv1, vin, perm_d, K);
    { int rc_ = apply_refined(pv1, st); if (rc_) return rc_; }
    hipLaunchKernelGGL(ts_perm_kernel<false>, dim3((K + 255) / 256), dim3(256), 0, st, vin, pv1, perm_d, K);
    CUADMM_HIP_TRY(hipGetLastError());
    return CUADMM_OK;
  }
  // this rank's rows r = K - 1 - i (r = 0: the longest row): equal shares of the triangle's entries, boundaries on multiples of 8
  int r_begin = 0, r_end = K;
  if (shard_world > 1) {
    r_begin = tail_shard_bound(K, shard_rank, shard_world);
    r_end = tail_shard_bound(K, shard_rank + 1, shard_world);
  }
  if (compact && (r_begin != K - 1 - i_hi || r_end != K - i_lo)) {
    set_error("tail_solve: this rank keeps rows %d .. %d of inv(L22) only (keep_shard) but was asked for another share", i_lo, i_hi);
    return CUADMM_ERR_INVALID;
  }
  shard_rows = r_end - r_begin;
  shard_bytes = 8.0 * ((double)(r_end - r_begin) * (double)K - 0.5 * ((double)r_end * r_end - (double)r_begin * r_begin));   // entries of the rows read
  { int rc_ = apply_rows(st, r_begin, r_end); if (rc_) return rc_; }
  if (shard_world > 1) {
    if (!reduce_fn) { set_error("tail_solve: sharded over %d ranks but no reduction installed", shard_world); return CUADMM_ERR_COMM; }
    return reduce_fn(reduce_user, vin, (size_t)K, st);      // x = sum over ranks of W_r^T D_r^-1 W_r z: identical on every rank afterwards
  }
  return CUADMM_OK;
}

int TailSolve::apply_rows(hipStream_t st, int r_begin, int r_end) {
  const long long ldw = compact ? ldc : (long long)K;      // leading dimension of W (a compact shard keeps its rows at their own width)
  const int nc = (K + 1023) / 1024;
  const size_t lds = sizeof(double) * 1024 * (size_t)(nc + (nc & 1));
  // z as the producer left it: in the caller's order (gathered through perm_d while it is staged) or already in the factor's (z_scatter())
  const bool z_pivot = vin_pivot;
  vin_pivot = false;
  const int* perm_in = z_pivot ? nullptr : perm_d;
  if (z_pivot && !linear_z_ok()) { set_error("tail_solve: z was written in the factor's order but the pass that reads it that way is gone"); return CUADMM_ERR_INVALID; }
  if (one_pass && xpart && lds <= kMaxLdsBytes - 1024 && nc <= 20) {
    auto launch = [&](auto kern) -> int {
      if (lds > 48 * 1024 && !attr_set) {      // once per object: K, and with it the instantiation, never changes
        CUADMM_HIP_TRY(allow_max_dynamic_lds(reinterpret_cast<const void*>(kern)));   // process-wide per kernel: the maximum
        attr_set = true;
      }
      hipLaunchKernelGGL(kern, dim3(n_wg), dim3(1024), lds, st, W, ldw, K, vin, dinv, xpart, r_begin, r_end, perm_in);
      return CUADMM_OK;
    };
    int rc;
    // K <= 10 240: eight fat wavefronts, two rows per group, two groups in flight (above ts_onepass_kernel); NC = columns per thread in steps of 4
    auto launch512 = [&](auto kern, int ncol) -> int {
      const size_t lds5 = sizeof(double) * 512 * (size_t)ncol;
      if (lds5 > 48 * 1024 && !attr_set) { CUADMM_HIP_TRY(allow_max_dynamic_lds(reinterpret_cast<const void*>(kern))); attr_set = true; }
      hipLaunchKernelGGL(kern, dim3(n_wg), dim3(512), lds5, st, W, ldw, K, vin, dinv, xpart, r_begin, r_end, perm_in);
      return CUADMM_OK;
    };
    if (prefetch && fat && !dd_dot && K <= 10240) {
      switch ((K + 2047) / 2048) {
        case 1: rc = launch512(ts_onepass_kernel<4, 2, 1, false, 512>, 4); break;
        case 2: rc = launch512(ts_onepass_kernel<8, 2, 1, false, 512>, 8); break;
        case 3: rc = launch512(ts_onepass_kernel<12, 2, 1, false, 512>, 12); break;
        case 4: rc = launch512(ts_onepass_kernel<16, 2, 1, false, 512>, 16); break;
        default: rc = launch512(ts_onepass_kernel<20, 2, 1, false, 512>, 20); break;
      }
    } else
    // NC columns per thread (1024 NC >= K), RB rows per group, D groups in flight beyond the current one during its barrier
    // ((D + 1) register buffers of RB NC doubles + NC accumulators within 128 VGPRs).  Option tail_depth: 0 = one group in flight everywhere
    // (rounds 3 - 5), 1 = one ahead (two rows per group up to NC = 8, one up to NC = 16, none beyond), 2 / 3 = two / three single rows ahead
    // where the registers allow (NC <= 12 / NC <= 8).  Option tail_order: the walk of ts_onepass_kernel.  Same column ownership everywhere;
    // the row grouping and the walk -- and with them the last bits -- differ between the settings.
    if (dd_dot && (nc == 9 || nc == 10)) rc = launch(ts_onepass_kernel<10, 1, 1, true>);      // experiment (option tail_dd): compensated u = W z
    else if (dd_dot && (nc == 15 || nc == 16)) rc = launch(ts_onepass_kernel<16, 1, 0, true>);
    else {
      // the requested grouping, degraded until the instantiation exists (ts_onepass_fits: the row buffers and accumulators within the budget)
      int rb_ = rows_per_group, dep = prefetch ? depth : 0;
      if (rb_ != 1 && rb_ != 2) rb_ = nc <= 8 ? 2 : 1;                    // rounds 3 - 6: two rows per group up to NC = 8
      rc = -1;
      while (rc == -1) {
        switch ((nc + 1) / 2) {
          case 1: rc = ts_onepass_launch<2>(rb_, dep, order, zreg, launch); break;
          case 2: rc = ts_onepass_launch<4>(rb_, dep, order, zreg, launch); break;
          case 3: rc = ts_onepass_launch<6>(rb_, dep, order, zreg, launch); break;
          case 4: rc = ts_onepass_launch<8>(rb_, dep, order, zreg, launch); break;
          case 5: rc = ts_onepass_launch<10>(rb_, dep, order, zreg, launch); break;
          case 6: rc = ts_onepass_launch<12>(rb_, dep, order, zreg, launch); break;
          case 7: rc = ts_onepass_launch<14>(rb_, dep, order, zreg, launch); break;
          case 8: rc = ts_onepass_launch<16>(rb_, dep, order, zreg, launch); break;
          case 9: rc = ts_onepass_launch<18>(rb_, dep, order, zreg, launch); break;
          default: rc = ts_onepass_launch<20>(rb_, dep, order, zreg, launch); break;
        }
        if (rc == -1) { if (dep > 0) --dep; else rb_ = 1; }              // (one row, nothing ahead: exists for every NC)
      }
    }
    if (rc) return rc;
    hipLaunchKernelGGL(ts_onepass_reduce_kernel, dim3((K + 31) / 32), dim3(256), 0, st, xpart, K, n_wg, vin, (unsigned long long*)nullptr, 0, perm_d);
  } else if (one_pass && xpart && part && K > 32768 && K <= 65536 && !group_retired) {
    // beyond 32 768 columns (round 5, option tail_max_k): EIGHT workgroups share a row, 8 columns per thread, four rows per exchange
    constexpr int Q = 8;
    const size_t lds2 = sizeof(double) * 1024 * 8;
    const int G = std::max(8, 2 * n_wg / Q / 8 * 8);
    auto kern = ts_onepass_group_kernel<8, Q, 4, 4>;
    if (!attr_set) { CUADMM_HIP_TRY(allow_max_dynamic_lds(reinterpret_cast<const void*>(kern))); attr_set = true; }
    hipLaunchKernelGGL(kern, dim3(G * Q), dim3(1024), lds2, st, W, ldw, K, vin, dinv, xpart, part, d_fail, r_begin, r_end, perm_in, order);
    hipLaunchKernelGGL(ts_onepass_reduce_kernel, dim3((K + 31) / 32), dim3(256), 0, st, xpart, K, G, vin, part, Q, perm_d);
  } else if (one_pass && xpart && part && K <= 32768 && !group_retired) {
    constexpr int Q = 4;
    // measured (tail_solve class per sGS iteration, two solves; two triangular GEMVs for comparison): K = 24 576 (PushBox N = 30, forced)
    // 1.99 -> 1.36 ms with 6 columns per thread, two rows per exchange, two workgroups per CU; K = 27 136 (PushT_N=30) 2.83 -> 2.10 and
    // K = 30 720 (PushBox N = 50) 3.1 -> 2.16 with 8 columns per thread, FOUR rows per exchange and one workgroup per CU (128 VGPRs) --
    // one row per exchange at 64 VGPRs: 2.24 / 2.27; two rows spill; eight members of 4 columns: 2.67 (the group waits for its slowest)
    const bool small = K <= 1024 * Q * 6;
    const size_t lds2 = sizeof(double) * 1024 * (size_t)(small ? 6 : 8);
    const int G = std::max(8, 2 * n_wg / Q / 8 * 8);            // groups: two workgroups per CU, whole octets (member m of a group: + 8 m)
    auto launch = [&](auto kern) -> int {
      if (lds2 > 48 * 1024 && !attr_set) { CUADMM_HIP_TRY(allow_max_dynamic_lds(reinterpret_cast<const void*>(kern))); attr_set = true; }
      hipLaunchKernelGGL(kern, dim3(G * Q), dim3(1024), lds2, st, W, ldw, K, vin, dinv, xpart, part, d_fail, r_begin, r_end, perm_in, order);
      return CUADMM_OK;
    };
    // (group_pf: two rows per exchange and the next group's rows in flight across it, instead of four rows and nothing in flight)
    int rc = small ? launch(ts_onepass_group_kernel<6, Q, 2, 8>) : group_pf ? launch(ts_onepass_group_kernel<8, Q, 2, 4, true>) : launch(ts_onepass_group_kernel<8, Q, 4, 4>);
    if (rc) return rc;
    hipLaunchKernelGGL(ts_onepass_reduce_kernel, dim3((K + 31) / 32), dim3(256), 0, st, xpart, K, G, vin, part, Q, perm_d);
  } else {
    // (with a pivoted factor the vector goes into the factor's order first and back afterwards: two short launches on this fallback only)
    const double* zin = vin;
    double* xout = vin;
    if (perm_d) {
      hipLaunchKernelGGL(ts_perm_kernel<true>, dim3((K + 255) / 256), dim3(256), 0, st, pv1, vin, perm_d, K);
      zin = pv1; xout = pv2;
    }
    hipLaunchKernelGGL(ts_tri_gemv_kernel<true>, dim3((K + 3) / 4), dim3(256), 0, st, W, ldw, K, zin, dinv, vmid, r_begin, r_end);
    if (compact) {     // no W^T on a compact shard: the kept rows are accumulated by columns (chunks of rows, summed in chunk order)
      const int nchunk = std::max(1, std::min(n_wg, 64));
      if (i_hi >= i_lo) hipLaunchKernelGGL(ts_colacc_kernel, dim3((K + 255) / 256, nchunk), dim3(256), 0, st, W, ldw, K, vmid, xpart, i_lo, i_hi, nchunk);
      else CUADMM_HIP_TRY(hipMemsetAsync(xpart, 0, sizeof(double) * (size_t)K * (size_t)nchunk, st));
      hipLaunchKernelGGL(ts_onepass_reduce_kernel, dim3((K + 31) / 32), dim3(256), 0, st, xpart, K, nchunk, vin, (unsigned long long*)nullptr, 0, perm_d);
    } else {
      hipLaunchKernelGGL(ts_tri_gemv_kernel<false>, dim3((K + 3) / 4), dim3(256), 0, st, Wt, (long long)K, K, vmid, nullptr, xout, 0, 1 << 30);
      if (perm_d) hipLaunchKernelGGL(ts_perm_kernel<false>, dim3((K + 255) / 256), dim3(256), 0, st, vin, pv2, perm_d, K);
    }
  }
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

}  // namespace cuadmm
