// HBM-bound kernels of the ADMM iteration (gfx950): fused vector updates, the two SpMVs,
// deterministic reductions, plus the op-level kernels that mirror the reference's own
// element-wise CUDA kernels one to one (used by the parity tests through the C ABI).
//
// Fusion map (reference call sites in src/solver.cu):
//   aty_xb     : SpMV At*y (:514) + D2D copy (:518) + Rd1 -= C (:520) + Xb = X + sig*Rd1 (:527)
//   post_admm  : Xdiff (:652) + S (:656) + Rd (:746) + X update (:758) + ||Rd||^2 (:775-776)
//                + <C,X> (:774), one pass
//   post_S / post_X : the same split around the second sGS solve (:693-729)
//   spmv_rows  : -A*SmC (:478,:695) and -A*X (:764) in one pass over A (SmC = S - C formed on the fly,
//                :672-674 removed)
// All reductions are two-stage (per-workgroup partials, then one workgroup) => run-to-run
// deterministic, unlike atomics.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "device_util.h"
#include "vec_kernels.h"
#include "psd_fuse.h"

namespace cuadmm {

constexpr int kVecThreads = 256;

static inline int grid_for(long long n, int per_block, int cap = 256 * 8) {
  long long g = (n + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}

// block-wide sums of up to 2 values; result valid in thread 0
template <int NT>
__device__ __forceinline__ void block_sum2(double& a, double& b) {
  __shared__ double sa[NT / 64], sb[NT / 64];
  a = wave_sum(a);
  b = wave_sum(b);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sa[w] = a; sb[w] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ta = 0, tb = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) { ta += sa[i]; tb += sb[i]; }
    a = ta; b = tb;
  }
}

// ------------------------------------------------------------------------------------------
// Rd1 = At*y - C ; Xb = X + sig*Rd1          (one thread per svec row; rows are mostly 0..few nnz)
// ------------------------------------------------------------------------------------------
// Rows with more than `skip_above` entries (> 0) are left to aty_xb_long_kernel: a moment relaxation has svec slots --
// the (1,1) entry of the moment matrix -- that appear in thousands of constraints (PushT_N=10: one row with 2720 of
// the 46 388 nonzeros made this kernel take 0.39 ms of a 0.89 ms iteration).
// FOUR rows per thread and pass (rows i, i + stride, ...: every access still coalesced across the wavefront): the row pointers, C
// and X of all four are in flight before the first data-dependent nonzero loop starts.  With one row per thread the kernel was
// bound by memory-level parallelism, not bandwidth: 3.3 TB/s at BASELINE configs[3] size (41 % of 8 TB/s).
template <bool WRITE_XB>
__global__ __launch_bounds__(kVecThreads) void aty_xb_kernel(long long L, const int* __restrict__ rp,
                                                             const int* __restrict__ ci, const double* __restrict__ av,
                                                             const double* __restrict__ y, const double* __restrict__ C,
                                                             const double* __restrict__ X, double sig,
                                                             double* __restrict__ Rd1, double* __restrict__ Xb, int skip_above) {
  constexpr int R = 4;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i0 < L; i0 += R * stride) {
    int p0[R], p1[R];
    double c[R], x[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const long long i = i0 + k * stride;
      const long long ii = i < L ? i : i0;                   // out of range: the thread's first row again (masked below)
      p0[k] = rp[ii]; p1[k] = rp[ii + 1]; c[k] = C[ii];
      x[k] = WRITE_XB ? X[ii] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const long long i = i0 + k * stride;
      if (i >= L || (skip_above > 0 && p1[k] - p0[k] > skip_above)) continue;
      double t = 0.0;
      for (int p = p0[k]; p < p1[k]; ++p) t += av[p] * y[ci[p]];
      const double r = t - c[k];
      Rd1[i] = r;
      if (WRITE_XB) Xb[i] = x[k] + r * sig;
    }
  }
}

// EIGHT lanes per row: a moment relaxation has svec rows of up to the long-row cap (128) entries among thousands of short ones
// (PlanarHand_N=1: mean 2.8, 7.7 % of the rows above 8), and one thread per row walks such a row one dependent gather after the
// other -- the longest row decides: 52 us for 55 k rows, latency, not bandwidth.  Lane s of a row takes entries s, s + 8, ...; three
// shuffles add the partial sums in a fixed order.  Taken when a row below the cap is longer than 24.
template <bool WRITE_XB>
__global__ __launch_bounds__(kVecThreads) void aty_xb_g8_kernel(long long L, const int* __restrict__ rp, const int* __restrict__ ci,
                                                                const double* __restrict__ av, const double* __restrict__ y,
                                                                const double* __restrict__ C, const double* __restrict__ X, double sig,
                                                                double* __restrict__ Rd1, double* __restrict__ Xb, int skip_above) {
  const long long gt = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long i = gt >> 3;
  const int sub = (int)(gt & 7);
  const bool in = i < L;
  const int p0 = in ? rp[i] : 0, p1 = in ? rp[i + 1] : 0;
  const bool skip = skip_above > 0 && p1 - p0 > skip_above;
  double t = 0.0;
  if (!skip)
    for (int p = p0 + sub; p < p1; p += 8) t += av[p] * y[ci[p]];
  t += __shfl_xor(t, 4, 64);
  t += __shfl_xor(t, 2, 64);
  t += __shfl_xor(t, 1, 64);
  if (in && !skip && sub == 0) {
    const double r = t - C[i];
    Rd1[i] = r;
    if (WRITE_XB) Xb[i] = X[i] + r * sig;
  }
}

// the same over a LIST of svec rows (fused iteration: the rows outside the blocks whose projection kernel does this itself)
__global__ __launch_bounds__(kVecThreads) void aty_xb_idx_kernel(long long nidx, const int* __restrict__ idx, const int* __restrict__ rp,
                                                                 const int* __restrict__ ci, const double* __restrict__ av,
                                                                 const double* __restrict__ y, const double* __restrict__ C,
                                                                 const double* __restrict__ X, double sig,
                                                                 double* __restrict__ Rd1, double* __restrict__ Xb) {
  for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nidx; k += (long long)gridDim.x * blockDim.x) {
    const int i = idx[k];
    double t = 0.0;
    for (int p = rp[i]; p < rp[i + 1]; ++p) t += av[p] * y[ci[p]];
    const double r = t - C[i];
    Rd1[i] = r;
    Xb[i] = X[i] + r * sig;
  }
}

// Second half of an sGS iteration in ONE pass (solver.cu:707-729,746-758,774-776): Rd1 = A^T y - C with the new y, Rd = Rd1 + S,
// X += tau sigma Rd, sum Rd^2, <C, X> -- aty_xb_kernel<false> followed by post_kernel<2> wrote and re-read Rd1 (16 B per svec
// element of 76) and took two launches.  Same expressions; Rd1 is not stored (the next iteration forms its own).
__global__ __launch_bounds__(kVecThreads) void aty_post2_kernel(long long L, const int* __restrict__ rp, const int* __restrict__ ci,
                                                                const double* __restrict__ av, const double* __restrict__ y,
                                                                const double* __restrict__ C, const double* __restrict__ S,
                                                                double* __restrict__ X, double tau_sig, double* __restrict__ partials) {
  double s_rd = 0.0, s_cx = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (long long)gridDim.x * blockDim.x) {
    double t = 0.0;
    for (int p = rp[i]; p < rp[i + 1]; ++p) t += av[p] * y[ci[p]];
    const double c = C[i];
    const double r1 = t - c;
    const double rd = r1 + S[i];
    const double xn = X[i] + tau_sig * rd;
    X[i] = xn;
    s_rd += rd * rd;
    s_cx += c * xn;
  }
  block_sum2<kVecThreads>(s_rd, s_cx);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = s_rd; partials[2 * blockIdx.x + 1] = s_cx; }
}

// one workgroup per long row (two-stage sum in a fixed order: reproducible)
template <bool WRITE_XB>
__global__ __launch_bounds__(kVecThreads) void aty_xb_long_kernel(const int* __restrict__ long_rows, const int* __restrict__ rp,
                                                                  const int* __restrict__ ci, const double* __restrict__ av,
                                                                  const double* __restrict__ y, const double* __restrict__ C,
                                                                  const double* __restrict__ X, double sig,
                                                                  double* __restrict__ Rd1, double* __restrict__ Xb) {
  __shared__ double red[kVecThreads];
  const int i = long_rows[blockIdx.x];
  double t = 0.0;
  for (int p = rp[i] + (int)threadIdx.x; p < rp[i + 1]; p += kVecThreads) t += av[p] * y[ci[p]];
  red[threadIdx.x] = t;
  __syncthreads();
  for (int o = kVecThreads / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double r = red[0] - C[i];
    Rd1[i] = r;
    if (WRITE_XB) Xb[i] = X[i] + r * sig;
  }
}

// ------------------------------------------------------------------------------------------
// after the projection
// ------------------------------------------------------------------------------------------
// mode 0 (ADMM): S = (Xproj - X)/sig - Rd1 ; Rd = Rd1 + S ; X += tau*sig*Rd ; sums
// mode 1 (sGS first half): S only
// mode 2 (sGS second half): Rd = Rd1 + S ; X += tau*sig*Rd ; sums        (S read, not written)
template <int MODE>
__global__ __launch_bounds__(kVecThreads) void post_kernel(long long L, const double* __restrict__ Xproj,
                                                           const double* __restrict__ Rd1, const double* __restrict__ C,
                                                           double* __restrict__ X, double* __restrict__ S,
                                                           double inv_sig, double tau_sig, double* __restrict__ partials) {
  double s_rd = 0.0, s_cx = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (long long)gridDim.x * blockDim.x) {
    const double x = X[i];
    const double r1 = Rd1[i];
    double s;
    if (MODE == 2) {
      s = S[i];
    } else {
      const double xdiff = Xproj[i] - x;
      s = inv_sig * xdiff - r1;
      S[i] = s;
    }
    if (MODE != 1) {
      const double rd = r1 + s;
      const double xn = x + tau_sig * rd;
      X[i] = xn;
      s_rd += rd * rd;
      s_cx += C[i] * xn;
    }
  }
  if (MODE != 1) {
    block_sum2<kVecThreads>(s_rd, s_cx);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = s_rd; partials[2 * blockIdx.x + 1] = s_cx; }
  }
}

// post_kernel<0|1> over a LIST of svec rows (fused iteration); partial sums at partials[2 * blockIdx.x]
template <int MODE>
__global__ __launch_bounds__(kVecThreads) void post_idx_kernel(long long nidx, const int* __restrict__ idx, const double* __restrict__ Xproj,
                                                               const double* __restrict__ Rd1, const double* __restrict__ C,
                                                               double* __restrict__ X, double* __restrict__ S,
                                                               double inv_sig, double tau_sig, double* __restrict__ partials) {
  double s_rd = 0.0, s_cx = 0.0;
  for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nidx; k += (long long)gridDim.x * blockDim.x) {
    const int i = idx[k];
    const double x = X[i];
    const double r1 = Rd1[i];
    const double xdiff = Xproj[i] - x;
    const double s = inv_sig * xdiff - r1;
    S[i] = s;
    if (MODE == 0) {
      const double rd = r1 + s;
      const double xn = x + tau_sig * rd;
      X[i] = xn;
      s_rd += rd * rd;
      s_cx += C[i] * xn;
    }
  }
  if (MODE == 0) {
    block_sum2<kVecThreads>(s_rd, s_cx);
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = s_rd; partials[2 * blockIdx.x + 1] = s_cx; }
  }
}

// sums `nparts` (a,b) pairs into out[0], out[1]: one workgroup of 1024 threads, four independent partial sums per thread
// (the fused iteration leaves one pair per block: 10 000 pairs took 13 us with 256 threads and one dependent chain each);
// fixed assignment and order: bit-reproducible
constexpr int kReduceThreads = 1024;
__global__ __launch_bounds__(kReduceThreads) void reduce_pairs_kernel(const double* __restrict__ partials, int nparts,
                                                                      double* __restrict__ out) {
  double a0 = 0.0, b0 = 0.0, a1 = 0.0, b1 = 0.0, a2 = 0.0, b2 = 0.0, a3 = 0.0, b3 = 0.0;
  const double2* __restrict__ pp = reinterpret_cast<const double2*>(partials);
  int i = threadIdx.x;
  for (; i + 3 * kReduceThreads < nparts; i += 4 * kReduceThreads) {
    const double2 v0 = pp[i], v1 = pp[i + kReduceThreads], v2 = pp[i + 2 * kReduceThreads], v3 = pp[i + 3 * kReduceThreads];
    a0 += v0.x; b0 += v0.y; a1 += v1.x; b1 += v1.y; a2 += v2.x; b2 += v2.y; a3 += v3.x; b3 += v3.y;
  }
  for (; i < nparts; i += kReduceThreads) { const double2 v = pp[i]; a0 += v.x; b0 += v.y; }
  double a = (a0 + a1) + (a2 + a3), b = (b0 + b1) + (b2 + b3);
  block_sum2<kReduceThreads>(a, b);
  if (threadIdx.x == 0) { out[0] = a; out[1] = b; }
}

// Fused iteration with closed blocks (psd_fuse.h): the projection kernels left BOTH partial pairs per block -- (sum Rd^2,
// <C, X>) in p1 (followed by the pairs of the stand-alone post step) and (||Rp org||^2, b^T y) in p2 -- and this one
// workgroup forms all four scalars of the stopping test: out4 = [||Rp||^2, b.y, sum Rd^2, <C,X>], sums_out = out4[2..3].
// The sums are formed per SEGMENT of kQuadSeg consecutive pairs (one workgroup per segment: thread t adds pairs t, t + 1024, ... of
// its segment into two alternating accumulators, then the workgroup's tree) and the segment sums are added in segment order.  Up to
// kQuadSeg pairs (BASELINE configs[1]: 10 000 blocks) that is the single-workgroup reduction of round 2, bit for bit; beyond (configs[3]:
// 100 000 blocks, 57 us through one CU) the segments run on as many workgroups.  Every variant -- one launch per iteration, several
// iterations per launch -- uses this association: the same bits.
constexpr int kQuadSeg = 16384;
__device__ __forceinline__ void quad_segment_sum(const double2* __restrict__ q, int n, int seg, double& a, double& b) {
  const int beg = seg * kQuadSeg, end = n < beg + kQuadSeg ? n : beg + kQuadSeg;
  double a0 = 0.0, b0 = 0.0, a1 = 0.0, b1 = 0.0;
  int i = beg + (int)threadIdx.x;
  for (; i + kReduceThreads < end; i += 2 * kReduceThreads) { const double2 u = q[i], v = q[i + kReduceThreads]; a0 += u.x; b0 += u.y; a1 += v.x; b1 += v.y; }
  for (; i < end; i += kReduceThreads) { const double2 u = q[i]; a0 += u.x; b0 += u.y; }
  a = a0 + a1; b = b0 + b1;
  block_sum2<kReduceThreads>(a, b);      // valid in thread 0
  __syncthreads();
}
__global__ __launch_bounds__(kReduceThreads) void reduce_quads_kernel(const double* __restrict__ p1, int n1, const double* __restrict__ p2, int n2,
                                                                     double* __restrict__ out4, double* __restrict__ sums_out, double* __restrict__ seg_out) {
  const double2* __restrict__ q1 = reinterpret_cast<const double2*>(p1);
  const double2* __restrict__ q2 = reinterpret_cast<const double2*>(p2);
  const int sg = (int)blockIdx.x;
  double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
  if (sg * kQuadSeg < n1 || sg == 0) quad_segment_sum(q1, n1, sg, a, b);
  if (sg * kQuadSeg < n2 || sg == 0) quad_segment_sum(q2, n2, sg, c, d);
  if (threadIdx.x == 0) {
    if (gridDim.x == 1) { out4[0] = c; out4[1] = d; out4[2] = a; out4[3] = b; sums_out[0] = a; sums_out[1] = b; }
    else { double* o = seg_out + 4 * (size_t)sg; o[0] = c; o[1] = d; o[2] = a; o[3] = b; }
  }
}
__global__ void reduce_quads_final_kernel(const double* __restrict__ seg_out, int nseg, double* __restrict__ out4, double* __restrict__ sums_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double v[4] = {seg_out[0], seg_out[1], seg_out[2], seg_out[3]};
  for (int s = 1; s < nseg; ++s)
    for (int k = 0; k < 4; ++k) v[k] += seg_out[4 * (size_t)s + k];
  for (int k = 0; k < 4; ++k) out4[k] = v[k];
  sums_out[0] = v[2]; sums_out[1] = v[3];
}

// Several iterations per launch (SignFuse::iters): workgroup k forms the four scalars of iteration k from that iteration's partial
// arrays (p1 + k stride, p2 + k stride; n pairs each) with the assignment and order of reduce_quads_kernel -- the same bits as
// one launch per iteration.
__global__ __launch_bounds__(kReduceThreads) void reduce_quads_batch_kernel(const double* __restrict__ p1, const double* __restrict__ p2, int n,
                                                                           long long stride, double* __restrict__ out4) {
  const double2* __restrict__ q1 = reinterpret_cast<const double2*>(p1 + (long long)blockIdx.x * stride);
  const double2* __restrict__ q2 = reinterpret_cast<const double2*>(p2 + (long long)blockIdx.x * stride);
  const int nseg = n > 0 ? (n + kQuadSeg - 1) / kQuadSeg : 1;
  double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
  for (int sg = 0; sg < nseg; ++sg) {
    double sa, sb, sc, sd;
    quad_segment_sum(q1, n, sg, sa, sb);
    quad_segment_sum(q2, n, sg, sc, sd);
    if (sg == 0) { a = sa; b = sb; c = sc; d = sd; } else { a += sa; b += sb; c += sc; d += sd; }
  }
  if (threadIdx.x == 0) { double* o = out4 + 4 * (size_t)blockIdx.x; o[0] = c; o[1] = d; o[2] = a; o[3] = b; }
}

int launch_reduce_quads_batch(const double* p1, const double* p2, int n, long long stride, int iters, double* out4, hipStream_t st) {
  hipLaunchKernelGGL(reduce_quads_batch_kernel, dim3(iters), dim3(kReduceThreads), 0, st, p1, p2, n, stride, out4);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

// closed blocks keep a per-block copy of their rows of [A X | A (S - C)] (psd_sign_closed.h); after a stand-alone kernel wrote
// those vectors by row (init, restart, the sGS half step's A X) the copy is refreshed from them
__global__ __launch_bounds__(kVecThreads) void closed_gather_out_kernel(const ClosedRec* __restrict__ rec, int nslots, const double* __restrict__ ax,
                                                                      const double* __restrict__ as, double* __restrict__ cl_out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int slot = (int)(i / kClosedMaxRows), k = (int)(i % kClosedMaxRows);
  if (slot >= nslots || k >= rec[slot].nk) return;
  const int row = rec[slot].rows[k];
  cl_out[16 * (long long)slot + k] = ax[row];
  cl_out[16 * (long long)slot + 8 + k] = as[row];
}
int launch_closed_gather_out(const ClosedRec* rec, int nslots, const double* ax, const double* as, double* cl_out, hipStream_t st) {
  if (nslots <= 0) return CUADMM_OK;
  const long long n = (long long)nslots * kClosedMaxRows;
  hipLaunchKernelGGL(closed_gather_out_kernel, dim3((unsigned)((n + kVecThreads - 1) / kVecThreads)), dim3(kVecThreads), 0, st, rec, nslots, ax, as, cl_out);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

int reduce_quads_segments(int n1, int n2) {
  const int nmax = n1 > n2 ? n1 : n2;
  return nmax > 0 ? (nmax + kQuadSeg - 1) / kQuadSeg : 1;
}
// seg_scratch: 4 doubles per segment (reduce_quads_segments), only touched when there is more than one segment
int launch_reduce_quads(const double* p1, int n1, const double* p2, int n2, double* out4, double* sums_out, double* seg_scratch, hipStream_t st) {
  const int nseg = reduce_quads_segments(n1, n2);
  if (nseg > 1 && !seg_scratch) { set_error("reduce_quads: no scratch for %d segments", nseg); return CUADMM_ERR_INVALID; }
  hipLaunchKernelGGL(reduce_quads_kernel, dim3(nseg), dim3(kReduceThreads), 0, st, p1, n1, p2, n2, out4, sums_out, seg_scratch);
  if (nseg > 1) hipLaunchKernelGGL(reduce_quads_final_kernel, dim3(1), dim3(64), 0, st, seg_scratch, nseg, out4, sums_out);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

// Owned-constraints sharding: the two constraint-space sums of the stopping test, formed on the device from A*X so that
// all four scalars of an iteration travel in ONE small all-reduce enqueued on the stream (no host round trip):
//   out4[0] = sum_i (normA_i (b_i - (A X)_i) bscale)^2   (|| Rp org ||^2, solver.cu:768-772)
//   out4[1] = sum_i b_i y_i                               (solver.cu:781)
//   out4[2..3] = sums[0..1] (sum Rd^2, <C, X>: copied next to them)
// Two stages over a fixed grid, fixed summation order: bit-reproducible.
__global__ __launch_bounds__(kVecThreads) void rp_stats_partial_kernel(int m, const double* __restrict__ ax, const double* __restrict__ b,
                                                                       const double* __restrict__ normA, const double* __restrict__ y,
                                                                       double bscale, double* __restrict__ partials) {
  double a = 0.0, c = 0.0;
  for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < m; i += (int)(gridDim.x * blockDim.x)) {
    const double ro = normA[i] * (b[i] - ax[i]) * bscale;
    a += ro * ro;
    c += b[i] * y[i];
  }
  block_sum2<kVecThreads>(a, c);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = a; partials[2 * blockIdx.x + 1] = c; }
}
__global__ __launch_bounds__(kVecThreads) void rp_stats_final_kernel(const double* __restrict__ partials, int nparts, const double* __restrict__ sums,
                                                                     double* __restrict__ out4) {
  double a = 0.0, c = 0.0;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) { a += partials[2 * i]; c += partials[2 * i + 1]; }
  block_sum2<kVecThreads>(a, c);
  if (threadIdx.x == 0) { out4[0] = a; out4[1] = c; out4[2] = sums[0]; out4[3] = sums[1]; }
}
// two-stage, fixed grid (64 workgroups) and fixed summation order: bit-reproducible; `partials` holds 128 doubles
int launch_rp_stats(int m, const double* ax, const double* b, const double* normA, const double* y, double bscale,
                    const double* sums, double* partials, double* out4, hipStream_t st) {
  constexpr int kGrid = 64;
  hipLaunchKernelGGL(rp_stats_partial_kernel, dim3(kGrid), dim3(kVecThreads), 0, st, m, ax, b, normA, y, bscale, partials);
  hipLaunchKernelGGL(rp_stats_final_kernel, dim3(1), dim3(kVecThreads), 0, st, partials, kGrid, sums, out4);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

// ------------------------------------------------------------------------------------------
// The y-solve of a block-diagonal A A^T on the device (engine option: automatic when the elimination forest of the factor
// consists of many small trees): y = (L D L^T)^-1 rhs with rhs = -A(S-C) + (b - A X) / sigma (solver.cu:478-500), one
// THREAD per tree.  The sweeps of a solve never leave a tree, and inside a tree this is the serial host algorithm
// (aat_ldlt.cpp: columns ascending, x[i] -= L[i][j] x[j]; then x[j] / D[j] and the transposed sweep) with unfused
// multiply-subtract, so y is bit-identical to the host solve.  y stays in HBM: no m-vector crosses PCIe in an iteration.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kVecThreads) void forest_solve_kernel(int ntrees, const int* __restrict__ tree_ptr, const int* __restrict__ tree_cols,
                                                                   const long long* __restrict__ Lp, const int* __restrict__ Li,
                                                                   const double* __restrict__ Lx, const double* __restrict__ D,
                                                                   const double* __restrict__ ax, const double* __restrict__ asmc,
                                                                   const double* __restrict__ b, double isig, double* __restrict__ x) {
  const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (t >= ntrees) return;
  const int c0 = tree_ptr[t], c1 = tree_ptr[t + 1];
  for (int q = c0; q < c1; ++q) {
    const int j = tree_cols[q];
    const double rp = __dadd_rn(-ax[j], b[j]);                       // Rp = -A X + b
    x[j] = __dadd_rn(-asmc[j], __dmul_rn(isig, rp));                 // rhs = -A(S-C) + Rp / sigma
  }
  for (int q = c0; q < c1; ++q) {                                    // L z = rhs
    const int j = tree_cols[q];
    const double xj = x[j];
    if (xj != 0.0)
      for (long long p = Lp[j]; p < Lp[j + 1]; ++p) x[Li[p]] = __dsub_rn(x[Li[p]], __dmul_rn(Lx[p], xj));
  }
  for (int q = c1 - 1; q >= c0; --q) {                               // D^-1, then L^T y = z
    const int j = tree_cols[q];
    double s = x[j] / D[j];
    for (long long p = Lp[j]; p < Lp[j + 1]; ++p) s = __dsub_rn(s, __dmul_rn(Lx[p], x[Li[p]]));
    x[j] = s;
  }
}
int launch_forest_solve(int ntrees, const int* tree_ptr, const int* tree_cols, const long long* Lp, const int* Li, const double* Lx,
                        const double* D, const double* ax, const double* asmc, const double* b, double isig, double* x, hipStream_t st) {
  if (ntrees <= 0) return CUADMM_OK;
  hipLaunchKernelGGL(forest_solve_kernel, dim3((ntrees + kVecThreads - 1) / kVecThreads), dim3(kVecThreads), 0, st, ntrees, tree_ptr, tree_cols, Lp,
                     Li, Lx, D, ax, asmc, b, isig, x);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

// ------------------------------------------------------------------------------------------
// rows of A (constraints, already in the solver's permuted order): T lanes per row.
//   outX[row]  = sum a * X[col]            (if outX)
//   outS[row]  = sum a * (S[col] - C[col]) (if outS)
// ------------------------------------------------------------------------------------------
// `cap` > 0: only the first `cap` nonzeros of a row are summed here; the rest of such LONG rows (a trace or all-ones
// constraint can hold as many nonzeros as the whole svec) is cut into segments summed by spmv_segments_kernel and
// added in segment order by spmv_finish_kernel -- one row must not serialise the launch (swissroll: one row with
// 320 000 of the 330 537 nonzeros made this kernel take 4.8 ms).
template <int T>
__global__ __launch_bounds__(kVecThreads) void spmv_rows_kernel(int rows, const int* __restrict__ rp,
                                                                const int* __restrict__ ci, const double* __restrict__ av,
                                                                const double* __restrict__ X, const double* __restrict__ S,
                                                                const double* __restrict__ C, double* __restrict__ outX,
                                                                double* __restrict__ outS, int cap, const int* __restrict__ rowmap) {
  const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int sub = (int)(threadIdx.x & (T - 1));
  const long long nsub = (long long)gridDim.x * blockDim.x / T;
  const bool doX = outX != nullptr, doS = outS != nullptr;
  for (long long row = gtid / T; row < rows; row += nsub) {
    const int p0 = rp[row];
    int p1 = rp[row + 1];
    if (cap > 0 && p1 - p0 > cap) p1 = p0 + cap;
    double ax = 0.0, as = 0.0;
    for (int p = p0 + sub; p < p1; p += T) {
      const int c = ci[p];
      const double a = av[p];
      if (doX) ax += a * X[c];
      if (doS) as += a * (S[c] - C[c]);
    }
#pragma unroll
    for (int o = T >> 1; o > 0; o >>= 1) {
      ax += __shfl_xor(ax, o, 64);
      as += __shfl_xor(as, o, 64);
    }
    if (sub == 0) {
      const long long orow = rowmap ? rowmap[row] : row;     // compact row list (fused iteration): the constraint's own slot
      if (doX) outX[orow] = ax;
      if (doS) outS[orow] = as;
    }
  }
}

// one workgroup per segment [seg_begin, seg_end) of a long row: partial[2*seg] (A X), partial[2*seg+1] (A (S - C))
__global__ __launch_bounds__(kVecThreads) void spmv_segments_kernel(const int* __restrict__ seg_begin, const int* __restrict__ seg_end,
                                                                    const int* __restrict__ ci, const double* __restrict__ av,
                                                                    const double* __restrict__ X, const double* __restrict__ S,
                                                                    const double* __restrict__ C, bool doX, bool doS,
                                                                    double* __restrict__ partial) {
  __shared__ double rx[kVecThreads], rs[kVecThreads];
  const int p0 = seg_begin[blockIdx.x], p1 = seg_end[blockIdx.x];
  double ax = 0.0, as = 0.0;
  for (int p = p0 + (int)threadIdx.x; p < p1; p += kVecThreads) {
    const int c = ci[p];
    const double a = av[p];
    if (doX) ax += a * X[c];
    if (doS) as += a * (S[c] - C[c]);
  }
  rx[threadIdx.x] = ax; rs[threadIdx.x] = as;
  __syncthreads();
  for (int o = kVecThreads / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { rx[threadIdx.x] += rx[threadIdx.x + o]; rs[threadIdx.x] += rs[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = rx[0]; partial[2 * blockIdx.x + 1] = rs[0]; }
}
// out[row] += its segments' partials, in segment order (one thread per long row)
__global__ void spmv_finish_kernel(int nlong, const int* __restrict__ long_row, const int* __restrict__ long_seg0,
                                   const double* __restrict__ partial, double* __restrict__ outX, double* __restrict__ outS) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= nlong) return;
  double ax = 0.0, as = 0.0;
  for (int sgi = long_seg0[i]; sgi < long_seg0[i + 1]; ++sgi) { ax += partial[2 * sgi]; as += partial[2 * sgi + 1]; }
  if (outX) outX[long_row[i]] += ax;
  if (outS) outS[long_row[i]] += as;
}

int launch_aty_xb(bool write_xb, long long L, const int* rp, const int* ci, const double* av, const double* y,
                  const double* C, const double* X, double sig, double* Rd1, double* Xb, hipStream_t st, const AtyLongRows* lr) {
  const int grid = grid_for((L + 3) / 4, kVecThreads, 256 * 16);
  const int skip = (lr && lr->nlong > 0) ? lr->cap : 0;
  if (lr && lr->max_short > 24) {                       // rows of dozens of entries below the long-row cap: eight lanes per row
    const long long g8 = (L * 8 + kVecThreads - 1) / kVecThreads;
    if (write_xb) hipLaunchKernelGGL(aty_xb_g8_kernel<true>, dim3((unsigned)g8), dim3(kVecThreads), 0, st, L, rp, ci, av, y, C, X, sig, Rd1, Xb, skip);
    else hipLaunchKernelGGL(aty_xb_g8_kernel<false>, dim3((unsigned)g8), dim3(kVecThreads), 0, st, L, rp, ci, av, y, C, X, sig, Rd1, Xb, skip);
  } else if (write_xb)
    hipLaunchKernelGGL(aty_xb_kernel<true>, dim3(grid), dim3(kVecThreads), 0, st, L, rp, ci, av, y, C, X, sig, Rd1, Xb, skip);
  else
    hipLaunchKernelGGL(aty_xb_kernel<false>, dim3(grid), dim3(kVecThreads), 0, st, L, rp, ci, av, y, C, X, sig, Rd1, Xb, skip);
  if (skip > 0) {
    if (write_xb)
      hipLaunchKernelGGL(aty_xb_long_kernel<true>, dim3(lr->nlong), dim3(kVecThreads), 0, st, lr->rows, rp, ci, av, y, C, X, sig, Rd1, Xb);
    else
      hipLaunchKernelGGL(aty_xb_long_kernel<false>, dim3(lr->nlong), dim3(kVecThreads), 0, st, lr->rows, rp, ci, av, y, C, X, sig, Rd1, Xb);
  }
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

int launch_aty_xb_idx(long long nidx, const int* idx, const int* rp, const int* ci, const double* av, const double* y, const double* C,
                      const double* X, double sig, double* Rd1, double* Xb, hipStream_t st) {
  if (nidx <= 0) return CUADMM_OK;
  hipLaunchKernelGGL(aty_xb_idx_kernel, dim3(grid_for(nidx, kVecThreads, 256 * 16)), dim3(kVecThreads), 0, st, nidx, idx, rp, ci, av, y, C, X, sig, Rd1, Xb);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

int launch_aty_post2(long long L, const int* rp, const int* ci, const double* av, const double* y, const double* C, const double* S, double* X,
                     double tau_sig, double* partials, double* sums_out, hipStream_t st) {
  const int grid = post_grid(L);
  hipLaunchKernelGGL(aty_post2_kernel, dim3(grid), dim3(kVecThreads), 0, st, L, rp, ci, av, y, C, S, X, tau_sig, partials);
  hipLaunchKernelGGL(reduce_pairs_kernel, dim3(1), dim3(kReduceThreads), 0, st, partials, grid, sums_out);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

int AtyLongRows::build(long long L, const int* rp_host) {
  release();
  std::vector<int> lr;
  for (long long i = 0; i < L; ++i)
    if (rp_host[i + 1] - rp_host[i] > cap) lr.push_back((int)i);
  max_short = 0;
  for (long long i = 0; i < L; ++i) { const int len = rp_host[i + 1] - rp_host[i]; if (len <= cap && len > max_short) max_short = len; }
  nlong = (int)lr.size();
  if (nlong == 0) return CUADMM_OK;
  CUADMM_HIP_TRY(hipMalloc(&rows, sizeof(int) * lr.size()));
  { int rc_ = staged_h2d(rows, lr.data(), sizeof(int) * lr.size()); if (rc_) return rc_; }
  return CUADMM_OK;
}
void AtyLongRows::release() {
  if (rows) { hipError_t e = hipFree(rows); (void)e; }
  rows = nullptr;
  nlong = 0;
}

int post_grid(long long L) { return grid_for(L, kVecThreads * 4, 256 * 8); }

int launch_post(int mode, long long L, const double* Xproj, const double* Rd1, const double* C, double* X, double* S,
                double inv_sig, double tau_sig, double* partials, double* sums_out, hipStream_t st) {
  const int grid = post_grid(L);
  switch (mode) {
    case 0: hipLaunchKernelGGL(post_kernel<0>, dim3(grid), dim3(kVecThreads), 0, st, L, Xproj, Rd1, C, X, S, inv_sig, tau_sig, partials); break;
    case 1: hipLaunchKernelGGL(post_kernel<1>, dim3(grid), dim3(kVecThreads), 0, st, L, Xproj, Rd1, C, X, S, inv_sig, tau_sig, partials); break;
    default: hipLaunchKernelGGL(post_kernel<2>, dim3(grid), dim3(kVecThreads), 0, st, L, Xproj, Rd1, C, X, S, inv_sig, tau_sig, partials); break;
  }
  CUADMM_HIP_TRY(hipGetLastError());
  if (mode != 1) {
    hipLaunchKernelGGL(reduce_pairs_kernel, dim3(1), dim3(kReduceThreads), 0, st, partials, grid, sums_out);
    CUADMM_HIP_TRY(hipGetLastError());
  }
  return CUADMM_OK;
}

// Fused iteration: the projection kernels left one partial pair per fused block in partials[0, 2 nfused); the rows of `idx`
// add their grid's pairs behind them, and (mode 0) everything is summed in slot order.
int launch_post_rest(int mode, long long nidx, const int* idx, int nfused, const double* Xproj, const double* Rd1, const double* C, double* X,
                     double* S, double inv_sig, double tau_sig, double* partials, double* sums_out, hipStream_t st, int* nparts_out) {
  const int grid = nidx > 0 ? post_grid(nidx) : 0;
  if (nidx > 0) {
    double* part = partials + 2 * (size_t)nfused;
    if (mode == 0) hipLaunchKernelGGL(post_idx_kernel<0>, dim3(grid), dim3(kVecThreads), 0, st, nidx, idx, Xproj, Rd1, C, X, S, inv_sig, tau_sig, part);
    else hipLaunchKernelGGL(post_idx_kernel<1>, dim3(grid), dim3(kVecThreads), 0, st, nidx, idx, Xproj, Rd1, C, X, S, inv_sig, tau_sig, part);
    CUADMM_HIP_TRY(hipGetLastError());
  }
  if (nparts_out) { *nparts_out = nfused + grid; return CUADMM_OK; }    // the caller reduces (launch_reduce_quads)
  if (mode == 0) {
    hipLaunchKernelGGL(reduce_pairs_kernel, dim3(1), dim3(kReduceThreads), 0, st, partials, nfused + grid, sums_out);
    CUADMM_HIP_TRY(hipGetLastError());
  }
  return CUADMM_OK;
}

int launch_spmv_rows(int rows, double avg_nnz, const int* rp, const int* ci, const double* av, const double* X,
                     const double* S, const double* C, double* outX, double* outS, hipStream_t st, const SpmvLongRows* lr, const int* rowmap) {
  if (rows <= 0) return CUADMM_OK;
  const int cap = (lr && lr->nlong > 0) ? lr->cap : 0;
  int T = 1;
  while (T < 64 && T < avg_nnz) T <<= 1;
  const int grid = grid_for((long long)rows * T, kVecThreads, 256 * 16);
#define CUADMM_SPMV_CASE(TT) \
  case TT: hipLaunchKernelGGL(spmv_rows_kernel<TT>, dim3(grid), dim3(kVecThreads), 0, st, rows, rp, ci, av, X, S, C, outX, outS, cap, rowmap); break;
  switch (T) {
    CUADMM_SPMV_CASE(1) CUADMM_SPMV_CASE(2) CUADMM_SPMV_CASE(4) CUADMM_SPMV_CASE(8)
    CUADMM_SPMV_CASE(16) CUADMM_SPMV_CASE(32) CUADMM_SPMV_CASE(64)
  }
#undef CUADMM_SPMV_CASE
  CUADMM_HIP_TRY(hipGetLastError());
  if (cap > 0) {
    hipLaunchKernelGGL(spmv_segments_kernel, dim3(lr->nseg), dim3(kVecThreads), 0, st, lr->seg_begin, lr->seg_end, ci, av, X, S, C,
                       outX != nullptr, outS != nullptr, lr->partial);
    hipLaunchKernelGGL(spmv_finish_kernel, dim3((lr->nlong + 63) / 64), dim3(64), 0, st, lr->nlong, lr->long_row, lr->long_seg0, lr->partial,
                       outX, outS);
    CUADMM_HIP_TRY(hipGetLastError());
  }
  return CUADMM_OK;
}

// rows with more than `cap` nonzeros: segments of `seg_len` nonzeros beyond the first `cap`
int SpmvLongRows::build(int rows, const int* rp_host) {
  release();
  // "long" is relative: 8x the average row, at least 256 nonzeros (a matrix of uniformly long rows has no long rows)
  cap = (int)std::max<long long>(256, rows > 0 ? 8LL * rp_host[rows] / rows : 0);
  std::vector<int> lrow, lseg0{0}, sb, se;
  for (int r = 0; r < rows; ++r) {
    const int len = rp_host[r + 1] - rp_host[r];
    if (len <= cap) continue;
    lrow.push_back(r);
    for (int p = rp_host[r] + cap; p < rp_host[r + 1]; p += seg_len) { sb.push_back(p); se.push_back(std::min(p + seg_len, rp_host[r + 1])); }
    lseg0.push_back((int)sb.size());
  }
  nlong = (int)lrow.size();
  nseg = (int)sb.size();
  if (nlong == 0) return CUADMM_OK;
  CUADMM_HIP_TRY(hipMalloc(&long_row, sizeof(int) * lrow.size()));
  CUADMM_HIP_TRY(hipMalloc(&long_seg0, sizeof(int) * lseg0.size()));
  CUADMM_HIP_TRY(hipMalloc(&seg_begin, sizeof(int) * sb.size()));
  CUADMM_HIP_TRY(hipMalloc(&seg_end, sizeof(int) * se.size()));
  CUADMM_HIP_TRY(hipMalloc(&partial, sizeof(double) * 2 * sb.size()));
  { int rc_ = staged_h2d(long_row, lrow.data(), sizeof(int) * lrow.size()); if (rc_) return rc_; }
  { int rc_ = staged_h2d(long_seg0, lseg0.data(), sizeof(int) * lseg0.size()); if (rc_) return rc_; }
  { int rc_ = staged_h2d(seg_begin, sb.data(), sizeof(int) * sb.size()); if (rc_) return rc_; }
  { int rc_ = staged_h2d(seg_end, se.data(), sizeof(int) * se.size()); if (rc_) return rc_; }
  return CUADMM_OK;
}
void SpmvLongRows::release() {
  for (void* p : {(void*)long_row, (void*)long_seg0, (void*)seg_begin, (void*)seg_end, (void*)partial}) if (p) { hipError_t e = hipFree(p); (void)e; }
  long_row = long_seg0 = seg_begin = seg_end = nullptr;
  partial = nullptr;
  nlong = nseg = 0;
}

// ------------------------------------------------------------------------------------------
// small helpers used by the engine
// ------------------------------------------------------------------------------------------
__global__ void scale_kernel(double* v, long long n, double s) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) v[i] *= s;
}
// dst <- src (device to device), as a kernel on the caller's stream: the runtime's own device-to-device copy measured at 0.3 ms
// for 42 MB on this driver (a copy-engine path); 16 bytes per lane, two in flight
__global__ __launch_bounds__(kVecThreads) void copy_kernel(double* __restrict__ dst, const double* __restrict__ src, long long n) {
  const long long n2 = n / 2, stride = (long long)gridDim.x * blockDim.x;
  const double2* __restrict__ s2 = reinterpret_cast<const double2*>(src);
  double2* __restrict__ d2 = reinterpret_cast<double2*>(dst);
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + stride < n2; i += 2 * stride) { const double2 a = s2[i], b = s2[i + stride]; d2[i] = a; d2[i + stride] = b; }
  for (; i < n2; i += stride) d2[i] = s2[i];
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) dst[n - 1] = src[n - 1];
}
int launch_copy(double* dst, const double* src, long long n, hipStream_t st) {
  if (n <= 0) return CUADMM_OK;
  hipLaunchKernelGGL(copy_kernel, dim3(grid_for(n / 2 + 1, kVecThreads, 256 * 8)), dim3(kVecThreads), 0, st, dst, src, n);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

typedef unsigned sl_copy16 __attribute__((ext_vector_type(4)));
// several device-to-device copies in ONE launch (the checkpoint of a batch: X, S, y, [A X | sums | A (S - C)], the schedule hints):
// 16-byte words where the length allows, 4-byte words for the tail; pointers 16-byte aligned (hipMalloc)
__global__ __launch_bounds__(kVecThreads) void copy_multi_kernel(CopyJobs j) {
  const long long stride = (long long)gridDim.x * kVecThreads;
  for (int q = 0; q < j.count; ++q) {
    const sl_copy16* __restrict__ s = reinterpret_cast<const sl_copy16*>(j.src[q]);
    sl_copy16* __restrict__ d = reinterpret_cast<sl_copy16*>(j.dst[q]);
    const long long n16 = j.nbytes[q] >> 4;
    for (long long i = (long long)blockIdx.x * kVecThreads + threadIdx.x; i < n16; i += stride) d[i] = s[i];
    const int tail4 = (int)((j.nbytes[q] & 15) >> 2);
    if (blockIdx.x == 0 && (int)threadIdx.x < tail4)
      reinterpret_cast<unsigned*>(j.dst[q])[4 * n16 + threadIdx.x] = reinterpret_cast<const unsigned*>(j.src[q])[4 * n16 + threadIdx.x];
  }
}
int launch_copy_multi(const CopyJobs& jobs, hipStream_t st) {
  long long most = 0;
  for (int q = 0; q < jobs.count; ++q) {
    if (jobs.nbytes[q] & 3) { set_error("copy_multi: length %lld is not a multiple of 4 bytes", jobs.nbytes[q]); return CUADMM_ERR_INVALID; }
    most = std::max(most, jobs.nbytes[q] >> 4);
  }
  if (jobs.count <= 0 || most <= 0) {
    bool any = false;
    for (int q = 0; q < jobs.count; ++q) any = any || jobs.nbytes[q] > 0;
    if (!any) return CUADMM_OK;
  }
  hipLaunchKernelGGL(copy_multi_kernel, dim3(grid_for(most + 1, kVecThreads, 256 * 8)), dim3(kVecThreads), 0, st, jobs);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

int launch_scale(double* v, long long n, double s, hipStream_t st) {
  if (n <= 0) return CUADMM_OK;
  hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n, kVecThreads)), dim3(kVecThreads), 0, st, v, n, s);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

// ------------------------------------------------------------------------------------------
// op-level kernels: one per reference kernel (same element-wise semantics)
// ------------------------------------------------------------------------------------------
// src/kernels/vec_mat_conversion.cu:11-34
__global__ void v2m_kernel(const double* Xb, double* large_mat, double* small_mat, const int* mB, const int* m1,
                           const int* m2, int vec_len) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < vec_len) {
    const int a = m1[idx], b = m2[idx];
    const double v = (a == b) ? Xb[idx] : Xb[idx] * 0x1.6a09e667f3bcdp-1;
    double* dst = (mB[idx] == 0) ? large_mat : small_mat;
    dst[a] = v;
    dst[b] = v;
  }
}
// src/kernels/vec_mat_conversion.cu:36-57
__global__ void m2v_kernel(double* Xb, const double* large_mat, const double* small_mat, const int* mB, const int* m1,
                           const int* m2, int vec_len) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < vec_len) {
    const int a = m1[idx], b = m2[idx];
    const double* src = (mB[idx] == 0) ? large_mat : small_mat;
    const double v = src[a];
    Xb[idx] = (a == b) ? v : v * 0x1.6a09e667f3bccp+0;
  }
}
// src/kernels/dense_scalar.cu:41-47
__global__ void max_zero_kernel(double* w, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) w[i] = fmax(w[i], 0.0);
}
// src/kernels/diagonal_batch.cu:11-23 (column-major: scales column j of matrix k by w[k*n+j])
__global__ void mul_diag_kernel(double* out, const double* in, const double* w, int n, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < total) {
    const long long n2 = (long long)n * n;
    const long long k = idx / n2;
    const int col = (int)((idx - k * n2) / n);
    out[idx] = in[idx] * w[k * n + col];
  }
}
// src/kernels/permutation.cu:12-18 (scatter)
__global__ void permute_kernel(double* v1, const double* v2, const int* perm, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v1[perm[i]] = v2[i];
}
// src/kernels/sparse_matrix_norm.cu:11-31
__global__ void normA_kernel(const int* cp, double* vals, double* normA, int con_num) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < con_num) {
    double nrm = 0.0;
    for (int p = cp[j]; p < cp[j + 1]; ++p) nrm += vals[p] * vals[p];
    nrm = fmax(1.0, sqrt(nrm));
    normA[j] = nrm;
    for (int p = cp[j]; p < cp[j + 1]; ++p) vals[p] /= nrm;
  }
}
// src/kernels/dense_dense.cu:15-24 / :28-37
__global__ void axpby2_kernel(double* v1, const double* v2, double a, double b, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v1[i] = a * v1[i] + b * v2[i];
}
__global__ void axpby3_kernel(double* v1, const double* v2, const double* v3, double a, double b, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v1[i] = a * v2[i] + b * v3[i];
}
// general CSR SpMV y = alpha*A*x + beta*y (include/cuadmm/cusparse.h:70-83), one wavefront-quarter per row
__global__ void spmv_csr_kernel(int rows, const int* rp, const int* ci, const double* av, const double* x, double* y,
                                double alpha, double beta) {
  constexpr int T = 16;
  const long long gtid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int sub = (int)(threadIdx.x & (T - 1));
  const long long nsub = (long long)gridDim.x * blockDim.x / T;
  for (long long row = gtid / T; row < rows; row += nsub) {
    double acc = 0.0;
    for (int p = rp[row] + sub; p < rp[row + 1]; p += T) acc += av[p] * x[ci[p]];
#pragma unroll
    for (int o = T >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (sub == 0) y[row] = (beta == 0.0) ? alpha * acc : alpha * acc + beta * y[row];
  }
}
__global__ __launch_bounds__(kVecThreads) void sumsq_partial_kernel(const double* v, long long n, double* partials) {
  double a = 0.0, b = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) a += v[i] * v[i];
  block_sum2<kVecThreads>(a, b);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = a; partials[2 * blockIdx.x + 1] = 0.0; }
}

// P = T * V^T per n x n column-major matrix on the FP64 matrix cores
// (the reference's cublasDgemmStridedBatched(N,T), include/cuadmm/cublas.h:18-35).
// One wavefront per 16x16 tile of P; v_mfma_f64_16x16x4_f64: lane l feeds A[i=l&15][k=l>>4],
// B[k=l>>4][j=l&15]; D[row=(l>>4)+4*reg][col=l&15].
typedef double v4f64 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mul_trans_mfma_kernel(double* __restrict__ P, const double* __restrict__ T,
                                                             const double* __restrict__ V, int n, int count,
                                                             int tiles_per_dim) {
  const int wave = (int)((blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6);
  const int lane = (int)(threadIdx.x & 63);
  const int tiles = tiles_per_dim * tiles_per_dim;
  const int mat = wave / tiles;
  if (mat >= count) return;
  const int tile = wave - mat * tiles;
  const int ti = tile % tiles_per_dim, tj = tile / tiles_per_dim;
  const double* Tm = T + (long long)mat * n * n;
  const double* Vm = V + (long long)mat * n * n;
  double* Pm = P + (long long)mat * n * n;
  const int i = ti * 16 + (lane & 15);   // row of T feeding A
  const int j = tj * 16 + (lane & 15);   // row of V feeding B (B[k][j] = V[j][k])
  const int kk = lane >> 4;
  v4f64 acc = {0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < n; k0 += 4) {
    const int k = k0 + kk;
    const double a = (i < n && k < n) ? Tm[(long long)k * n + i] : 0.0;
    const double b = (j < n && k < n) ? Vm[(long long)k * n + j] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  const int col = tj * 16 + (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = ti * 16 + (lane >> 4) + 4 * r;
    if (row < n && col < n) Pm[(long long)col * n + row] = acc[r];
  }
}

}  // namespace cuadmm

using namespace cuadmm;

#define ST(s) ((hipStream_t)(s))
#define LAUNCH1D(kern, n, st, ...)                                                          \
  do {                                                                                      \
    long long _n = (n);                                                                     \
    if (_n > 0) {                                                                           \
      hipLaunchKernelGGL(kern, dim3((unsigned)((_n + 255) / 256)), dim3(256), 0, ST(st), __VA_ARGS__); \
      CUADMM_HIP_TRY(hipGetLastError());                                                    \
    }                                                                                       \
  } while (0)

extern "C" {

int cuadmm_op_vector_to_matrices(const double* Xb, double* large_mat, double* small_mat, const int* map_B,
                                 const int* map_M1, const int* map_M2, int vec_len, void* stream) {
  LAUNCH1D(v2m_kernel, vec_len, stream, Xb, large_mat, small_mat, map_B, map_M1, map_M2, vec_len);
  return CUADMM_OK;
}
int cuadmm_op_matrices_to_vector(double* Xb, const double* large_mat, const double* small_mat, const int* map_B,
                                 const int* map_M1, const int* map_M2, int vec_len, void* stream) {
  LAUNCH1D(m2v_kernel, vec_len, stream, Xb, large_mat, small_mat, map_B, map_M1, map_M2, vec_len);
  return CUADMM_OK;
}
int cuadmm_op_max_zero(double* w, int64_t n, void* stream) {
  LAUNCH1D(max_zero_kernel, n, stream, w, (long long)n);
  return CUADMM_OK;
}
int cuadmm_op_mul_diag_batch(double* out, const double* in, const double* w, int n, int count, void* stream) {
  const long long total = (long long)n * n * count;
  LAUNCH1D(mul_diag_kernel, total, stream, out, in, w, n, total);
  return CUADMM_OK;
}
int cuadmm_op_mul_trans_batch(double* P, const double* T, const double* V, int n, int count, void* stream) {
  if (n <= 0 || count <= 0) return CUADMM_OK;
  const int tpd = (n + 15) / 16;
  const long long waves = (long long)tpd * tpd * count;
  const long long blocks = (waves + 3) / 4;
  hipLaunchKernelGGL(mul_trans_mfma_kernel, dim3((unsigned)blocks), dim3(256), 0, ST(stream), P, T, V, n, count, tpd);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}
int cuadmm_op_permute(double* v1, const double* v2, const int* perm, int n, void* stream) {
  LAUNCH1D(permute_kernel, n, stream, v1, v2, perm, n);
  return CUADMM_OK;
}
int cuadmm_op_get_normA(const int* col_ptrs, double* vals, double* normA, int con_num, void* stream) {
  LAUNCH1D(normA_kernel, con_num, stream, col_ptrs, vals, normA, con_num);
  return CUADMM_OK;
}
int cuadmm_op_spmv_csr(int rows, const int* row_ptrs, const int* col_ids, const double* vals, const double* x, double* y,
                       double alpha, double beta, void* stream) {
  if (rows <= 0) return CUADMM_OK;
  const int grid = grid_for((long long)rows * 16, 256, 256 * 16);
  hipLaunchKernelGGL(spmv_csr_kernel, dim3(grid), dim3(256), 0, ST(stream), rows, row_ptrs, col_ids, vals, x, y, alpha, beta);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}
int cuadmm_op_axpby2(double* v1, const double* v2, double alpha, double beta, int64_t n, void* stream) {
  LAUNCH1D(axpby2_kernel, n, stream, v1, v2, alpha, beta, (long long)n);
  return CUADMM_OK;
}
int cuadmm_op_axpby3(double* v1, const double* v2, const double* v3, double alpha, double beta, int64_t n, void* stream) {
  LAUNCH1D(axpby3_kernel, n, stream, v1, v2, v3, alpha, beta, (long long)n);
  return CUADMM_OK;
}
int cuadmm_op_norm2(const double* v, int64_t n, double* host_out, void* stream) {
  if (!host_out) { set_error("norm2: null output"); return CUADMM_ERR_INVALID; }
  *host_out = 0.0;
  if (n <= 0) return CUADMM_OK;
  const int grid = grid_for(n, kVecThreads * 4, 1024);
  double* buf = nullptr;
  CUADMM_HIP_TRY(hipMalloc(&buf, sizeof(double) * (2 * (size_t)grid + 2)));
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(grid), dim3(kVecThreads), 0, ST(stream), v, (long long)n, buf);
  hipLaunchKernelGGL(reduce_pairs_kernel, dim3(1), dim3(kReduceThreads), 0, ST(stream), buf, grid, buf + 2 * grid);
  double h[2] = {0, 0};
  hipError_t e = hipMemcpyAsync(h, buf + 2 * grid, sizeof(h), hipMemcpyDeviceToHost, ST(stream));
  if (e == hipSuccess) e = hipStreamSynchronize(ST(stream));
  hipError_t e2 = hipFree(buf);
  (void)e2;
  CUADMM_HIP_TRY(e);
  *host_out = sqrt(h[0]);
  return CUADMM_OK;
}

}  // extern "C"
