// Device-side sweeps over the sparse leading columns of a split A A^T factor (see lead_solve.h).
//
// The reference solves with CHOLMOD on the host (include/cuadmm/cholesky_cpu.h:146-155) between a D2H and an H2D copy of
// the m-vector (src/solver.cu:489-499); round 1 kept the sparse leading columns there and moved the dense trailing
// triangle to the GPU.  On moment relaxations the host part became the largest share of an iteration (pendulum N=80:
// 1.5 of 2.4 ms).  The leading columns of such factors form a forest of a few thousand shallow trees (PlanarHand_N=1:
// 2365 trees, depth <= 66; pendulum: 2071 trees, depth <= 60), so the sweeps are done here, one WAVEFRONT per tree, level by
// level, in gather form (deterministic: every sum has a fixed order, no atomics).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <atomic>
#include <limits>
#include <new>
#include <numeric>
#include <vector>

#include "common.h"
#include "device_util.h"
#include "lead_solve.h"
#include "psd_device.h"
#include "tail_solve.h"
#include "wave_reduce.h"

namespace cuadmm {

namespace {

template <class T>
int to_device(T*& d, const std::vector<T>& h) {
  CUADMM_HIP_TRY(hipMalloc(&d, sizeof(T) * std::max<size_t>(h.size(), 1)));
  return h.empty() ? CUADMM_OK : staged_h2d(d, h.data(), sizeof(T) * h.size());
}

__device__ __forceinline__ double lead_rhs(const double* __restrict__ ax, const double* __restrict__ asmc, const double* __restrict__ b,
                                           double isig, int i) {
  return -asmc[i] + isig * (-ax[i] + b[i]);      // solver.cu:478-482
}

// the same with the row looked up (dense tree tops: the sweeps run in B | T | K order, the caller's vectors do not)
__device__ __forceinline__ double lead_rhs_at(const double* __restrict__ ax, const double* __restrict__ asmc, const double* __restrict__ b,
                                              double isig, const int* __restrict__ rid, int i) {
  return lead_rhs(ax, asmc, b, isig, rid ? rid[i] : i);
}

// sum over a group of G lanes (G a power of two <= 64, groups aligned)
__device__ __forceinline__ double group_sum(double s, int G) {
  for (int o = 1; o < G; o <<= 1) s += __shfl_xor(s, o, 64);
  return s;
}

// One wavefront per tree, level by level.  The tree's solution values live in LDS under LOCAL indices (position in the
// tree's processing order), so the only level-to-level dependency -- the gather x[col] -- is an LDS access; the index /
// value streams (ptr, ci, v) are stored in processing order and do not depend on x.  G = lvl_g[l] lanes share a row (from
// the level's mean row length: root levels have few long rows, leaf levels many empty ones).
// Dynamic LDS per workgroup: xs[max_nodes] doubles | s_off[max_levels + 1] ints | s_g[max_levels] ints.
//
// forward:  x[i] = rhs[i] - sum_{j < i} L11[i][j] x[j]
__global__ __launch_bounds__(256) void lead_copy_kernel(const double* __restrict__ src, double* __restrict__ dst, int n) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i < n) dst[i] = src[i];
}

__global__ __launch_bounds__(64) void lead_forward_kernel(const int* __restrict__ lvl_ptr, const int* __restrict__ lvl_off, const int* __restrict__ lvl_g,
                                                          const int* __restrict__ nodes, const long long* __restrict__ ptr, const int* __restrict__ ci,
                                                          const double* __restrict__ v, const double* __restrict__ ax, const double* __restrict__ asmc,
                                                          const double* __restrict__ b, double isig, double* __restrict__ x, int max_nodes,
                                                          int max_levels, const int* __restrict__ tree_ids, const int* __restrict__ rid) {
  extern __shared__ double lead_smem[];
  double* xs = lead_smem;
  int* s_off = reinterpret_cast<int*>(xs + max_nodes);
  int* s_g = s_off + max_levels + 1;
  const int t = tree_ids[blockIdx.x], lane = (int)threadIdx.x;
  const int l0 = lvl_ptr[t], nlev = lvl_ptr[t + 1] - 1 - l0;
  for (int l = lane; l <= nlev; l += 64) s_off[l] = lvl_off[l0 + l];
  for (int l = lane; l < nlev; l += 64) s_g[l] = lvl_g[l0 + l];
  wave_fence();
  const int first = s_off[0];
  for (int l = 0; l < nlev; ++l) {
    const int G = s_g[l], sub = lane & (G - 1), grp = lane / G, ngrp = 64 / G;
    const int beg = s_off[l], end = s_off[l + 1];
    for (int base = beg; base < end; base += ngrp) {       // uniform trip count: the shuffles need every lane
      const int idx = base + grp;
      double s = 0.0;
      if (idx < end)
        for (long long q = ptr[idx] + sub; q < ptr[idx + 1]; q += G) s += v[q] * xs[ci[q]];
      s = group_sum(s, G);
      if (idx < end && sub == 0) {
        const int i = nodes[idx];
        const double xi = lead_rhs_at(ax, asmc, b, isig, rid, i) - s;
        xs[idx - first] = xi;
        x[i] = xi;
      }
    }
    wave_fence();                                          // the next level reads xs written by this one
  }
}

// z2[i] = rhs[n1 + i] - sum_j L21[i][j] z1[j]  -> the tail's input vector (8 lanes per tail row, fixed summation order)
__global__ __launch_bounds__(256) void lead_tail_rhs_kernel(int k, int n1, const long long* __restrict__ rp, const int* __restrict__ ci,
                                                            const double* __restrict__ v, const double* __restrict__ ax, const double* __restrict__ asmc,
                                                            const double* __restrict__ b, double isig, const double* __restrict__ z1,
                                                            double* __restrict__ z2) {
  const int gt = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const int i = gt >> 3, sub = gt & 7;
  if (i >= k) return;
  double s = 0.0;
  for (long long q = rp[i] + sub; q < rp[i + 1]; q += 8) s += v[q] * z1[ci[q]];
  s += __shfl_xor(s, 4, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 1, 64);
  if (sub == 0) z2[i] = lead_rhs(ax, asmc, b, isig, n1 + i) - s;
}

// hybrid solve: z2[i] = z[n1 + i] - sum_j L21[i][j] z[j] with z = [z1 | rhs2] uploaded by the host; LANES lanes per tail row (64 where the
// rows are long: PlanarHand_N=10 has ~1 000 entries per tail row), fixed summation order
template <int LANES>
__global__ __launch_bounds__(256) void lead_tail_rhs_vec_kernel(int k, int n1, const long long* __restrict__ rp, const int* __restrict__ ci,
                                                                const double* __restrict__ v, const double* __restrict__ z, double* __restrict__ z2) {
  const long long gt = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int i = (int)(gt / LANES), sub = (int)(gt % LANES);
  if (i >= k) return;
  double s = 0.0;
  for (long long q = rp[i] + sub; q < rp[i + 1]; q += LANES) s += v[q] * z[ci[q]];
#pragma unroll
  for (int o = LANES / 2; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
  if (sub == 0) z2[i] = z[n1 + i] - s;
}

constexpr int kLongColumn = 128;
// w[j] = sum over the TAIL rows of column j:  L21[i][j] x2[i]   (8 lanes per leading column; independent of the sweeps).
// Columns with more than kLongColumn tail rows take a wavefront each: eight lanes walk a column of 1 783 entries (PlanarHand_N=10 below its
// tree tops) in 223 dependent gathers -- 110 of the kernel's 139 us were that one column's latency.  ONE launch for both kinds (round 6; two
// launches until then: 12.5 + 6.7 us on PlanarHand_N=1): workgroups [0, wg_short) walk all columns eight lanes each and skip the long ones,
// workgroups beyond take four long columns each.  Same sums in the same order as the two kernels.
__global__ __launch_bounds__(256) void lead_l21t_kernel(int n1, const long long* __restrict__ tp, const int* __restrict__ tr, const double* __restrict__ tv,
                                                        const double* __restrict__ x2, double* __restrict__ w, int wg_short, int n_long,
                                                        const int* __restrict__ cols, double* __restrict__ copy_dst, int copy_n,
                                                        const int* __restrict__ copy_map) {
  // the solved tail (with tree tops: [x_T | x_K], scattered to the caller's order through copy_map) into y on the way -- it used to be a
  // 5 us launch of its own behind the backward sweeps
  for (int q = (int)(blockIdx.x * blockDim.x + threadIdx.x); q < copy_n; q += (int)(gridDim.x * blockDim.x)) copy_dst[copy_map ? copy_map[q] : q] = x2[q];
  if ((int)blockIdx.x >= wg_short) {
    const int i = ((int)blockIdx.x - wg_short) * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
    if (i >= n_long) return;
    const int j = cols[i];
    double s = 0.0;
    for (long long q = tp[j] + lane; q < tp[j + 1]; q += 64) s += tv[q] * x2[tr[q]];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) w[j] = s;
    return;
  }
  const int gt = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const int j = gt >> 3, sub = gt & 7;
  if (j >= n1) return;
  if (tp[j + 1] - tp[j] > kLongColumn) return;               // a wavefront's, above
  double s = 0.0;
  for (long long q = tp[j] + sub; q < tp[j + 1]; q += 8) s += tv[q] * x2[tr[q]];
  s += __shfl_xor(s, 4, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 1, 64);
  if (sub == 0) w[j] = s;
}
// returns whether the copy x2[0, copy_n) -> copy_dst rode along (no leading columns: no launch)
static bool launch_l21t(int n1, const long long* tp, const int* tr, const double* tv, const double* x2, double* w, int n_long, const int* cols, hipStream_t st,
                        double* copy_dst = nullptr, int copy_n = 0, const int* copy_map = nullptr) {
  if (n1 <= 0) return false;
  const int wg_short = (int)(((long long)n1 * 8 + 255) / 256);
  hipLaunchKernelGGL(lead_l21t_kernel, dim3((unsigned)(wg_short + (n_long + 3) / 4)), dim3(256), 0, st, n1, tp, tr, tv, x2, w, wg_short, n_long, cols, copy_dst,
                     copy_dst ? copy_n : 0, copy_map);
  return copy_dst != nullptr;
}

// backward, levels root side first:  x[j] = x[j] / D[j] - w[j] - sum_{leading i > j} L11[i][j] x[i]
__global__ __launch_bounds__(64) void lead_backward_kernel(const int* __restrict__ lvl_ptr, const int* __restrict__ lvl_off, const int* __restrict__ lvl_g,
                                                           const int* __restrict__ nodes, const long long* __restrict__ ptr, const int* __restrict__ ci,
                                                           const double* __restrict__ v, const double* __restrict__ D, const double* __restrict__ w,
                                                           double* __restrict__ x, int max_nodes, int max_levels, const int* __restrict__ tree_ids,
                                                           const int* __restrict__ rid, double* __restrict__ yout) {
  extern __shared__ double lead_smem[];
  double* xs = lead_smem;
  int* s_off = reinterpret_cast<int*>(xs + max_nodes);
  int* s_g = s_off + max_levels + 1;
  const int t = tree_ids[blockIdx.x], lane = (int)threadIdx.x;
  const int l0 = lvl_ptr[t], nlev = lvl_ptr[t + 1] - 1 - l0;
  for (int l = lane; l <= nlev; l += 64) s_off[l] = lvl_off[l0 + l];
  for (int l = lane; l < nlev; l += 64) s_g[l] = lvl_g[l0 + l];
  wave_fence();
  const int first = s_off[0];
  for (int l = 0; l < nlev; ++l) {
    const int G = s_g[l], sub = lane & (G - 1), grp = lane / G, ngrp = 64 / G;
    const int beg = s_off[l], end = s_off[l + 1];
    for (int base = beg; base < end; base += ngrp) {
      const int idx = base + grp;
      double s = 0.0;
      if (idx < end)
        for (long long q = ptr[idx] + sub; q < ptr[idx + 1]; q += G) s += v[q] * xs[ci[q]];
      s = group_sum(s, G);
      if (idx < end && sub == 0) {
        const int j = nodes[idx];
        const double xj = x[j] / D[j] - w[j] - s;
        xs[idx - first] = xj;
        x[j] = xj;
        if (yout) yout[rid[j]] = xj;
      }
    }
    wave_fence();
  }
}

// The same sweeps with the tree's WHOLE index / value stream resident in LDS.  The streaming kernels above pay two dependent
// global-memory round trips per level (row pointers, then entries: ~2.5 us on the loaded chip), and the deepest tree decides:
// 61 levels = 158 / 198 us per sweep on pendulum N = 80 -- 45 % of its iteration.  A tree's stream is ONE contiguous chunk
// (slots are in processing order, trees contiguous: on average 105 entries, 3 914 at most), so the wavefront copies it in
// two coalesced passes -- row offsets, node ids, level table; then the entries and the gathered right-hand sides -- and every
// level after that touches LDS only.  Same gather order, same group sums: bit-identical to the streaming kernels.
// LDS: xs[cnt] | rhs[cnt] | sv[nnz] doubles, then sptr[cnt + 1] | sci[nnz] | snode[cnt] | s_off[nlev + 1] | s_g[nlev] ints.
// sum over aligned groups of 2^lg lanes on the DPP crossbar (quad_perm, row_half_mirror, row_mirror: no LDS traffic) up to 16
// lanes, ds_bpermute beyond; the same association order as group_sum (xor butterfly from offset 1 up)
__device__ __forceinline__ double group_sum_dpp(double s, int lg) {
  if (lg >= 1) s += sw_dpp<0xB1>(s);       // quad_perm [1,0,3,2]                                   = xor 1
  if (lg >= 2) s += sw_dpp<0x4E>(s);       // quad_perm [2,3,0,1]                                   = xor 2
  if (lg >= 3) s += sw_dpp<0x141>(s);      // row_half_mirror (quads are uniform by now)            = xor 4
  if (lg >= 4) s += sw_dpp<0x140>(s);      // row_mirror (halves of a row are uniform by now)       = xor 8
  if (lg >= 5) s += __shfl_xor(s, 16, 64);
  if (lg >= 6) s += __shfl_xor(s, 32, 64);
  return s;
}

struct LeadTreeDesc { int l0, nlev, first, cnt, nnz, pad; long long q0; };     // per tree and sweep: everything the kernel start needs in one load

// NW wavefronts per tree: one for the many small trees; four for the few big ones (rows of a level are independent: the
// wavefronts take them in turn and meet at a workgroup barrier per level; the copy-in runs four times as wide)
// The body of one tree: `tid` counts the NW wavefronts that share it (NW = 1: the wavefront's lane), lead_smem is the tree's own LDS.
template <bool BACKWARD, int NW>
__device__ __forceinline__ void lead_sweep_lds_body(const LeadTreeDesc d, double* __restrict__ lead_smem, const int tid, const int* __restrict__ lvl_off,
                                                    const int* __restrict__ lvl_g, const int* __restrict__ nodes, const long long* __restrict__ ptr,
                                                    const int* __restrict__ ci, const double* __restrict__ v,
                                                    const double* __restrict__ ax, const double* __restrict__ asmc, const double* __restrict__ b, double isig,
                                                    const double* __restrict__ D, const double* __restrict__ w, double* __restrict__ x,
                                                    const int* __restrict__ rid, double* __restrict__ yout = nullptr) {
  constexpr int NT = 64 * NW;
  const int lane = tid & 63, wave = tid >> 6;
  const int l0 = d.l0, nlev = d.nlev, first = d.first, cnt = d.cnt, nnz = d.nnz;
  const long long q0 = d.q0;
  double* xs = lead_smem;
  double* rhs = xs + cnt;
  double* sv = rhs + cnt;
  int* sptr = reinterpret_cast<int*>(sv + nnz);
  int* sci = sptr + cnt + 1;
  int* snode = sci + nnz;
  int* s_off = snode + cnt;
  int* s_g = s_off + nlev + 1;
  // Copy-in in TWO memory round trips.  Written as plain loops (load, store to LDS, next element) every iteration waits for its
  // own load: five loops of up to five iterations each cost ~35 us on a big tree -- most of the kernel.  So: every load that is
  // addressed from the descriptor alone is issued first (U elements per thread and array), then the gathers that need the node
  // ids, then the stores; what does not fit U elements per thread follows in plain loops.
  constexpr int U = 8, UV = 16;
  {
    int r_off[U], r_g[U], r_ptr[U], r_nd[U], r_c[UV];
    double r_v[UV], r_rhs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = tid + u * NT;
      r_off[u] = i <= nlev ? lvl_off[l0 + i] : 0;
      r_g[u] = i < nlev ? lvl_g[l0 + i] : 1;
      r_ptr[u] = i <= cnt ? (int)(ptr[first + i] - q0) : 0;
      r_nd[u] = i < cnt ? nodes[first + i] : 0;
    }
#pragma unroll
    for (int u = 0; u < UV; ++u) {
      const int q = tid + u * NT;
      r_v[u] = q < nnz ? v[q0 + q] : 0.0;
      r_c[u] = q < nnz ? ci[q0 + q] : 0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = tid + u * NT, nd = r_nd[u];
      r_rhs[u] = i < cnt ? (BACKWARD ? x[nd] / D[nd] - w[nd] : lead_rhs_at(ax, asmc, b, isig, rid, nd)) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = tid + u * NT;
      if (i <= nlev) s_off[i] = r_off[u] - first;
      if (i < nlev) s_g[i] = 31 - __clz(r_g[u]);             // log2 of the lanes per row
      if (i <= cnt) sptr[i] = r_ptr[u];
      if (i < cnt) { snode[i] = r_nd[u]; rhs[i] = r_rhs[u]; }
    }
#pragma unroll
    for (int u = 0; u < UV; ++u) {
      const int q = tid + u * NT;
      if (q < nnz) { sv[q] = r_v[u]; sci[q] = r_c[u]; }
    }
  }
  for (int l = tid + U * NT; l <= nlev; l += NT) s_off[l] = lvl_off[l0 + l] - first;
  for (int l = tid + U * NT; l < nlev; l += NT) { const int G = lvl_g[l0 + l]; s_g[l] = 31 - __clz(G); }
  for (int i = tid + U * NT; i <= cnt; i += NT) sptr[i] = (int)(ptr[first + i] - q0);
  for (int i = tid + U * NT; i < cnt; i += NT) {
    const int nd = nodes[first + i];
    snode[i] = nd;
    rhs[i] = BACKWARD ? x[nd] / D[nd] - w[nd] : lead_rhs_at(ax, asmc, b, isig, rid, nd);
  }
  for (int q = tid + UV * NT; q < nnz; q += NT) { sv[q] = v[q0 + q]; sci[q] = ci[q0 + q]; }
  if (NW > 1) __syncthreads(); else wave_fence();
  // (Measured and rejected: fetching the level table, the row bounds and the first entry of every row one level ahead -- the level
  // time of 1 180 ticks on a big tree is not the chain of LDS reads; it did not move.)
  int beg = s_off[0];
  for (int l = 0; l < nlev; ++l) {
    const int lg = s_g[l], G = 1 << lg, sub = lane & (G - 1), grp = lane >> lg, ngrp = 64 >> lg;
    const int end = s_off[l + 1];
    for (int base = beg + wave * ngrp; base < end; base += NW * ngrp) {       // uniform trip count inside a wavefront: the reductions need every lane
      const int idx = base + grp;
      double s = 0.0;
      if (idx < end)
        for (int q = sptr[idx] + sub; q < sptr[idx + 1]; q += G) s += sv[q] * xs[sci[q]];
      s = group_sum_dpp(s, lg);
      if (idx < end && sub == 0) xs[idx] = rhs[idx] - s;
    }
    beg = end;
    if (NW > 1) __syncthreads(); else wave_fence();        // the next level reads xs written by this one
  }
  // the solution leaves in one pass at the end
  // (yout: the backward sweep of a solve with tree tops writes straight into the caller's y, in the caller's order -- round 6; until then a
  // scatter kernel of its own behind the sweeps)
  if (yout) { for (int i = tid; i < cnt; i += NT) yout[rid[snode[i]]] = xs[i]; }
  else { for (int i = tid; i < cnt; i += NT) x[snode[i]] = xs[i]; }
}

template <bool BACKWARD, int NW>
__global__ __launch_bounds__(64 * NW) void lead_sweep_lds_kernel(const LeadTreeDesc* __restrict__ desc, const int* __restrict__ lvl_off,
                                                            const int* __restrict__ lvl_g, const int* __restrict__ nodes, const long long* __restrict__ ptr,
                                                            const int* __restrict__ ci, const double* __restrict__ v,
                                                            const double* __restrict__ ax, const double* __restrict__ asmc, const double* __restrict__ b, double isig,
                                                            const double* __restrict__ D, const double* __restrict__ w, double* __restrict__ x,
                                                            const int* __restrict__ rid, double* __restrict__ yout) {
  extern __shared__ double lead_smem[];
  lead_sweep_lds_body<BACKWARD, NW>(desc[blockIdx.x], lead_smem, (int)threadIdx.x, lvl_off, lvl_g, nodes, ptr, ci, v, ax, asmc, b, isig, D, w, x, rid, yout);
}

// The few big trees and the many small ones in ONE launch (round 5): workgroups [0, n_big) take a big tree each on four wavefronts,
// the others FOUR small trees, one per wavefront (each with its own slice of the LDS; a small tree never meets a workgroup barrier).
// Before: two launches on two streams between a fork and a join event -- each cross-stream wait costs ~10 us of idle queue (kernel
// trace of pendulum N = 80: 14 us in front of the kernel behind the join), twice per solve.
template <bool BACKWARD>
__global__ __launch_bounds__(256) void lead_sweep_merged_kernel(const LeadTreeDesc* __restrict__ desc_big, int n_big, const LeadTreeDesc* __restrict__ desc_small,
                                                             int n_small, int small_doubles, const int* __restrict__ lvl_off,
                                                             const int* __restrict__ lvl_g, const int* __restrict__ nodes, const long long* __restrict__ ptr,
                                                             const int* __restrict__ ci, const double* __restrict__ v,
                                                             const double* __restrict__ ax, const double* __restrict__ asmc, const double* __restrict__ b, double isig,
                                                             const double* __restrict__ D, const double* __restrict__ w, double* __restrict__ x,
                                                             const int* __restrict__ rid, double* __restrict__ yout) {
  extern __shared__ double lead_smem[];
  const int blk = (int)blockIdx.x;
  if (blk < n_big) {
    lead_sweep_lds_body<BACKWARD, 4>(desc_big[blk], lead_smem, (int)threadIdx.x, lvl_off, lvl_g, nodes, ptr, ci, v, ax, asmc, b, isig, D, w, x, rid, yout);
  } else {
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int t = 4 * (blk - n_big) + wave;
    if (t < n_small)
      lead_sweep_lds_body<BACKWARD, 1>(desc_small[t], lead_smem + (size_t)wave * small_doubles, (int)threadIdx.x & 63, lvl_off, lvl_g, nodes, ptr, ci, v, ax, asmc, b, isig,
                                       D, w, x, rid, yout);
  }
}

// Micro trees (lead_solve.h): one thread per tree of one or two nodes, slots in the sweeps' processing order.
template <bool BACKWARD>
__global__ __launch_bounds__(256) void lead_micro_kernel(int n_micro, const int* __restrict__ first, const int* __restrict__ cnt, const int* __restrict__ nodes,
                                                         const long long* __restrict__ ptr, const int* __restrict__ ci, const double* __restrict__ v,
                                                         const double* __restrict__ ax, const double* __restrict__ asmc, const double* __restrict__ b, double isig,
                                                         const double* __restrict__ D, const double* __restrict__ w, double* __restrict__ x,
                                                         const int* __restrict__ rid, double* __restrict__ yout) {
  const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (t >= n_micro) return;
  const int s0 = first[t];
  const int n0 = nodes[s0];
  const double x0 = BACKWARD ? x[n0] / D[n0] - w[n0] : lead_rhs_at(ax, asmc, b, isig, rid, n0);
  if (yout) yout[rid[n0]] = x0; else x[n0] = x0;
  if (cnt[t] > 1) {
    const int n1 = nodes[s0 + 1];
    double s = 0.0;
    for (long long q = ptr[s0 + 1]; q < ptr[s0 + 2]; ++q) s += v[q] * (ci[q] == 0 ? x0 : 0.0);      // (one entry: the other node)
    const double r1 = BACKWARD ? x[n1] / D[n1] - w[n1] : lead_rhs_at(ax, asmc, b, isig, rid, n1);
    if (yout) yout[rid[n1]] = r1 - s; else x[n1] = r1 - s;
  }
}

// ---- dense tree tops (lead_solve.h) -------------------------------------------------------------------------------------------------
// rows of the extended tail [T | K] over the B columns: zext[e] = rhs(row e) - sum_j L[e][j] z_B[j]   (LANES lanes per row, fixed order)
template <int LANES>
__global__ __launch_bounds__(256) void lead_ext_rhs_kernel(int kext, int n1, const long long* __restrict__ rp, const int* __restrict__ ci,
                                                           const double* __restrict__ v, const double* __restrict__ ax, const double* __restrict__ asmc,
                                                           const double* __restrict__ b, double isig, const int* __restrict__ rid,
                                                           const double* __restrict__ zB, double* __restrict__ zext) {
  const long long gt = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int e = (int)(gt / LANES), sub = (int)(gt % LANES);
  if (e >= kext) return;
  double s = 0.0;
  for (long long q = rp[e] + sub; q < rp[e + 1]; q += LANES) s += v[q] * zB[ci[q]];
#pragma unroll
  for (int o = LANES / 2; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
  if (sub == 0) zext[e] = lead_rhs_at(ax, asmc, b, isig, rid, n1 + e) - s;
}

// out[i] = sum_c vals[off[i] + c] * in[start + c], c < len: a packed row of W_T (start = beg[i], the row ends at i) or of W_T^T (start = i,
// len[i] entries); one wavefront per row, lanes stride the row (coalesced), xor butterfly: a fixed order
__global__ __launch_bounds__(256) void tops_gemv_kernel(int nrows, const long long* __restrict__ off, const int* __restrict__ beg, const int* __restrict__ len,
                                                        const double* __restrict__ vals, const double* __restrict__ in, double* __restrict__ out, int acc) {
  const int i = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
  if (i >= nrows) return;
  const int start = beg ? beg[i] : i, n = beg ? i - start + 1 : len[i];
  const double* __restrict__ row = vals + off[i];
  const double* __restrict__ x = in + start;
  double s0 = 0.0, s1 = 0.0;
  int c = lane;
  for (; c + 64 < n; c += 128) { s0 += row[c] * x[c]; s1 += row[c + 64] * x[c + 64]; }
  if (c < n) s0 += row[c] * x[c];
  double s = s0 + s1;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) out[i] = acc ? out[i] + s : s;
}

// residual of the unit triangular system of the tops: out[i] = rhs[i] - x[i] - sum_c T[i][c] x[c], T = the strict part of L_TT by rows
// (forward) or of L_TT^T by rows (backward: the CSC arrays); 32 lanes per row
__global__ __launch_bounds__(256) void tops_resid_kernel(int n, const long long* __restrict__ rp, const int* __restrict__ ci, const double* __restrict__ v,
                                                         const double* __restrict__ x, const double* __restrict__ rhs, double* __restrict__ out) {
  const int gt = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const int i = gt >> 5, sub = gt & 31;
  if (i >= n) return;
  double s = 0.0;
  for (long long q = rp[i] + sub; q < rp[i + 1]; q += 32) s += v[q] * x[ci[q]];
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
  if (sub == 0) out[i] = rhs[i] - x[i] - s;
}

// the tail's right-hand side: out[i] = zK[i] - sum_c L_KT[i][c] z_T[c]   (one wavefront per tail row: several hundred entries each)
__global__ __launch_bounds__(256) void tops_k_rhs_kernel(int kt, const long long* __restrict__ rp, const int* __restrict__ ci, const double* __restrict__ v,
                                                         const double* __restrict__ zT, const double* __restrict__ zK, double* __restrict__ out,
                                                         const int* __restrict__ pinv) {
  const int i = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
  if (i >= kt) return;
  double s = 0.0;
  for (long long q = rp[i] + lane; q < rp[i + 1]; q += 64) s += v[q] * zT[ci[q]];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) out[pinv ? pinv[i] : i] = zK[i] - s;      // (pinv: straight into the tail factor's pivoting order, TailSolve::z_scatter)
}

// u[c] = z_T[c] / D_T[c] - sum_i L_KT[i][c] x_K[i]   (one wavefront per T column); the workgroups behind the last column copy the solved
// tail beside x_T (xK_out: what w = [L_TB; L_KB]^T [x_T; x_K] reads as one vector) -- a launch of its own before
__global__ __launch_bounds__(256) void tops_u_kernel(int nT, const long long* __restrict__ cp, const int* __restrict__ ri, const double* __restrict__ v,
                                                     const double* __restrict__ xK, const double* __restrict__ zT, const double* __restrict__ DT,
                                                     double* __restrict__ u, double* __restrict__ xK_out, int kt) {
  const int nb_u = (nT + 3) / 4;
  if ((int)blockIdx.x >= nb_u) {
    const int i = ((int)blockIdx.x - nb_u) * 256 + (int)threadIdx.x;
    if (i < kt) xK_out[i] = xK[i];
    return;
  }
  const int c = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
  if (c >= nT) return;
  double s = 0.0;
  for (long long q = cp[c] + lane; q < cp[c + 1]; q += 64) s += v[q] * xK[ri[q]];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) u[c] = zT[c] / DT[c] - s;
}

// the solution back in the caller's order: y[rid[p]] = p < n1 ? xp[p] : xext[p - n1]
__global__ __launch_bounds__(256) void tops_scatter_kernel(int m, int n1, const int* __restrict__ rid, const double* __restrict__ xp, const double* __restrict__ xext,
                                                           double* __restrict__ y) {
  const int p = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (p < m) y[rid[p]] = p < n1 ? xp[p] : xext[p - n1];
}

}  // namespace

void LeadSolve::release() {
  for (void* p : {(void*)rp21, (void*)ci21, (void*)v21, (void*)fptr, (void*)fci, (void*)fv_, (void*)bptr, (void*)bci, (void*)bv_, (void*)tptr,
                  (void*)tri, (void*)tv_, (void*)D1, (void*)wvec, (void*)nodes_f, (void*)nodes_b, (void*)lvl_ptr_f, (void*)lvl_ptr_b, (void*)lvl_off_f,
                  (void*)lvl_off_b, (void*)lvl_g_f, (void*)lvl_g_b, (void*)desc_small_f, (void*)desc_small_b, (void*)desc_big_f, (void*)desc_big_b, (void*)trees_stream})
    if (p) { hipError_t e = hipFree(p); (void)e; }
  rp21 = fptr = bptr = tptr = nullptr; ci21 = fci = bci = tri = nullptr; v21 = fv_ = bv_ = tv_ = D1 = wvec = nullptr;
  nodes_f = nodes_b = lvl_ptr_f = lvl_ptr_b = lvl_off_f = lvl_off_b = lvl_g_f = lvl_g_b = nullptr;
  if (aux) { hipError_t e = hipStreamDestroy(aux); (void)e; e = hipEventDestroy(ev_fork); (void)e; e = hipEventDestroy(ev_join); (void)e; aux = nullptr; ev_fork = ev_join = nullptr; }
  desc_small_f = desc_small_b = desc_big_f = desc_big_b = nullptr; trees_stream = nullptr; n_small = n_big = n_stream = 0;
  for (void* p : {(void*)rid, (void*)xp, (void*)zext, (void*)xext, (void*)zT, (void*)uT, (void*)DT, (void*)Wf, (void*)Wb, (void*)wf_off, (void*)wb_off,
                  (void*)wf_beg, (void*)wb_len, (void*)kt_rp, (void*)tk_cp, (void*)kt_ci, (void*)tk_ri, (void*)kt_v, (void*)tk_v, (void*)tt_rp, (void*)tt_cp,
                  (void*)tt_ci, (void*)tt_ri, (void*)tt_v, (void*)tt_cv})
    if (p) { hipError_t e = hipFree(p); (void)e; }
  rid = wf_beg = wb_len = kt_ci = tk_ri = nullptr; xp = zext = xext = zT = uT = DT = Wf = Wb = kt_v = tk_v = nullptr;
  wf_off = wb_off = kt_rp = tk_cp = tt_rp = tt_cp = nullptr; tt_ci = tt_ri = nullptr; tt_v = tt_cv = nullptr;
  tops = false; nT = 0; k_tail = 0; tops_bytes = 0; tops_blocks = tops_max = 0;
  if (long_cols_d) { hipError_t e = hipFree(long_cols_d); (void)e; long_cols_d = nullptr; }
  n_long = 0;
  if (micro_first) { hipError_t e = hipFree(micro_first); (void)e; micro_first = nullptr; }
  if (micro_cnt) { hipError_t e = hipFree(micro_cnt); (void)e; micro_cnt = nullptr; }
  n_micro = 0;
  if (zfull) { hipError_t e = hipFree(zfull); (void)e; zfull = nullptr; }
  if (h_w) { hipError_t e = hipHostFree(h_w); (void)e; h_w = nullptr; }
  if (h_z) { hipError_t e = hipHostFree(h_z); (void)e; h_z = nullptr; }
  hybrid = false; nnz21 = 0;
  ready = false;
}

static int hybrid_buffers(LeadSolve& L) {
  CUADMM_HIP_TRY(hipMalloc(&L.zfull, sizeof(double) * (size_t)std::max(L.m, 1)));
  CUADMM_HIP_TRY(hipHostMalloc(&L.h_w, sizeof(double) * (size_t)std::max(L.n1, 1), hipHostMallocDefault));
  CUADMM_HIP_TRY(hipHostMalloc(&L.h_z, sizeof(double) * (size_t)std::max(L.m, 1), hipHostMallocDefault));
  if (!L.wvec) CUADMM_HIP_TRY(hipMalloc(&L.wvec, sizeof(double) * (size_t)std::max(L.n1, 1)));
  return CUADMM_OK;
}

bool LeadSolve::demote_to_hybrid() {
  if (tops) { release(); return false; }              // built in B | T | K order: not what the host's sweeps index
  if (!ready || !l21_pays() || !rp21 || !tptr) { release(); return false; }
  // keep rp21 / ci21 / v21, tptr / tri / tv_ and wvec; everything of the sweeps goes
  for (void* p : {(void*)fptr, (void*)fci, (void*)fv_, (void*)bptr, (void*)bci, (void*)bv_, (void*)D1, (void*)nodes_f, (void*)nodes_b, (void*)lvl_ptr_f,
                  (void*)lvl_ptr_b, (void*)lvl_off_f, (void*)lvl_off_b, (void*)lvl_g_f, (void*)lvl_g_b, (void*)desc_small_f, (void*)desc_small_b,
                  (void*)desc_big_f, (void*)desc_big_b, (void*)trees_stream})
    if (p) { hipError_t e = hipFree(p); (void)e; }
  fptr = bptr = nullptr; fci = bci = nullptr; fv_ = bv_ = D1 = nullptr;
  nodes_f = nodes_b = lvl_ptr_f = lvl_ptr_b = lvl_off_f = lvl_off_b = lvl_g_f = lvl_g_b = nullptr;
  desc_small_f = desc_small_b = desc_big_f = desc_big_b = nullptr; trees_stream = nullptr; n_small = n_big = n_stream = 0;
  ready = false;
  if (hybrid_buffers(*this)) { release(); return false; }
  hybrid = true;
  return true;
}

int LeadSolve::apply_l21(double* x, bool x_pinned, TailSolve& tail, hipStream_t st) {
  if (!hybrid) { set_error("lead_solve: hybrid mode not built"); return CUADMM_ERR_INVALID; }
  const double* src = x;
  if (!x_pinned) { std::copy(x, x + m, h_z); src = h_z; }          // the runtime never reads pageable caller memory (staging.hip)
  CUADMM_HIP_TRY(hipMemcpyAsync(zfull, src, sizeof(double) * (size_t)m, hipMemcpyHostToDevice, st));
  if (nnz21 >= 256ll * k)
    hipLaunchKernelGGL(lead_tail_rhs_vec_kernel<64>, dim3((unsigned)(((long long)k * 64 + 255) / 256)), dim3(256), 0, st, k, n1, rp21, ci21, v21, zfull, tail.vin);
  else
    hipLaunchKernelGGL(lead_tail_rhs_vec_kernel<8>, dim3((unsigned)(((long long)k * 8 + 255) / 256)), dim3(256), 0, st, k, n1, rp21, ci21, v21, zfull, tail.vin);
  CUADMM_HIP_TRY(hipGetLastError());
  int rc = tail.solve_device(st);
  if (rc) return rc;
  launch_l21t(n1, tptr, tri, tv_, tail.vin, wvec, n_long, long_cols_d, st);
  CUADMM_HIP_TRY(hipGetLastError());
  CUADMM_HIP_TRY(hipMemcpyAsync(h_w, wvec, sizeof(double) * (size_t)n1, hipMemcpyDeviceToHost, st));
  double* dst = x_pinned ? x + n1 : h_z + n1;
  CUADMM_HIP_TRY(hipMemcpyAsync(dst, tail.vin, sizeof(double) * (size_t)k, hipMemcpyDeviceToHost, st));
  CUADMM_HIP_TRY(hipStreamSynchronize(st));
  if (!x_pinned) std::copy(h_z + n1, h_z + m, x + n1);
  return CUADMM_OK;
}

int LeadSolve::build(int m_, int k_, const int64_t* Lp, const int* Li, const double* Lx, const double* D, bool allow_hybrid) {
  release();
  if (k_ > 0 && m_ - k_ > 0 && tops_level != 0) {
    int level = tops_level > 0 ? tops_level : 0;
    if (tops_level < 0) {
      // automatic: only where the plain sweeps cannot run at all (a tree beyond 6 144 nodes or 2 048 levels: build_core's limits)
      const int n1o = m_ - k_;
      std::vector<int> parent((size_t)n1o, -1), height((size_t)n1o, 0), cnt((size_t)n1o, 1);
      int deepest = 0, largest = 0;
      for (int j = 0; j < n1o; ++j) {
        int pmin = -1;
        for (long long p = Lp[j]; p < Lp[j + 1]; ++p) if (Li[p] < n1o && (pmin < 0 || Li[p] < pmin)) pmin = Li[p];
        parent[j] = pmin;
        if (pmin >= 0) { height[pmin] = std::max(height[pmin], height[j] + 1); cnt[pmin] += cnt[j]; }
        else { deepest = std::max(deepest, height[j] + 1); largest = std::max(largest, cnt[j]); }
      }
      if (largest > 6144 || deepest > 2048) level = 32;
    }
    if (level > 0) {
      int rc;
      try { rc = build_tops(m_, k_, Lp, Li, Lx, D, level); }
      catch (const std::bad_alloc&) { set_error("lead_solve: the dense tree tops do not fit in host memory"); rc = CUADMM_ERR_FACTOR; }
      if (rc == CUADMM_OK && ready) return CUADMM_OK;
      if (rc != CUADMM_OK && rc != CUADMM_ERR_FACTOR) return rc;
      release();                                       // the cut did not help (or an inverse failed its check): the plain paths decide
    }
  }
  return build_core(m_, k_, Lp, Li, Lx, D, allow_hybrid);
}

// experiment (option pinv_tol, NOTEBOOK.md "Round 6"): pivots below the tolerance in magnitude become
// +infinity -- the kernels divide by D, so the component along such a direction is dropped (1 / d := 0), here as in the tail's dinv
static std::vector<double> pinv_pivots(const double* D, int n, double tol) {
  std::vector<double> out(D, D + n);
  if (tol > 0.0)
    for (auto& d : out) if (std::fabs(d) < tol) d = std::numeric_limits<double>::infinity();
  return out;
}

int LeadSolve::build_core(int m_, int k_, const int64_t* Lp, const int* Li, const double* Lx, const double* D, bool allow_hybrid) {
  m = m_; k = k_; n1 = m - k;
  if (k <= 0 || n1 < 0) return CUADMM_OK;
  if (n1 == 0) {
    // the whole factor is the dense tail: rhs -> tail -> y, nothing to sweep (the tail-rhs kernel walks empty rows of L21)
    std::vector<long long> zero((size_t)k + 1, 0);
    CUADMM_HIP_TRY(hipMalloc(&rp21, sizeof(long long) * zero.size()));
    { int rc_ = staged_h2d(rp21, zero.data(), sizeof(long long) * zero.size()); if (rc_) return rc_; }
    est_us = 20.0;
    ready = true;
    return CUADMM_OK;
  }
  const long long nnz = (long long)Lp[n1];
  // CSR of L11 and of L21 by counting sort over the leading columns (rows ascending inside a column => columns ascending inside a row)
  std::vector<long long> r11((size_t)n1 + 1, 0), r21((size_t)k + 1, 0);
  for (long long p = 0; p < nnz; ++p) { const int i = Li[p]; if (i < n1) r11[(size_t)i + 1]++; else r21[(size_t)(i - n1) + 1]++; }
  for (int i = 0; i < n1; ++i) r11[(size_t)i + 1] += r11[i];
  for (int i = 0; i < k; ++i) r21[(size_t)i + 1] += r21[i];
  std::vector<int> c11((size_t)r11[n1]), c21((size_t)r21[k]);
  std::vector<double> w11((size_t)r11[n1]), w21((size_t)r21[k]);
  {
    std::vector<long long> f11(r11.begin(), r11.end() - 1), f21(r21.begin(), r21.end() - 1);
    for (int j = 0; j < n1; ++j)
      for (long long p = Lp[j]; p < Lp[j + 1]; ++p) {
        const int i = Li[p];
        if (i < n1) { const long long q = f11[i]++; c11[(size_t)q] = j; w11[(size_t)q] = Lx[p]; }
        else { const long long q = f21[i - n1]++; c21[(size_t)q] = j; w21[(size_t)q] = Lx[p]; }
      }
  }
  // elimination forest of the leading block, levels of both sweeps
  std::vector<int> parent((size_t)n1, -1), root((size_t)n1), lev_f((size_t)n1, 0), lev_b((size_t)n1, 0);
  for (int j = 0; j < n1; ++j) {
    int pmin = -1;
    for (long long p = Lp[j]; p < Lp[j + 1]; ++p) if (Li[p] < n1 && (pmin < 0 || Li[p] < pmin)) pmin = Li[p];
    parent[j] = pmin;
  }
  for (int j = n1 - 1; j >= 0; --j) root[j] = parent[j] < 0 ? j : root[parent[j]];
  for (int i = 0; i < n1; ++i) {          // forward: a row waits for every column it gathers from
    int lv = 0;
    for (long long q = r11[i]; q < r11[i + 1]; ++q) lv = std::max(lv, lev_f[c11[(size_t)q]] + 1);
    lev_f[i] = lv;
  }
  for (int j = n1 - 1; j >= 0; --j) {     // backward: a column waits for every leading row it gathers from
    int lv = 0;
    for (long long p = Lp[j]; p < Lp[j + 1]; ++p) if (Li[p] < n1) lv = std::max(lv, lev_b[Li[p]] + 1);
    lev_b[j] = lv;
  }
  std::vector<int> tree_of((size_t)n1, -1);
  ntrees = 0;
  for (int j = 0; j < n1; ++j) if (root[j] == j) tree_of[j] = ntrees++;
  for (int j = 0; j < n1; ++j) tree_of[j] = tree_of[root[j]];
  auto order = [&](const std::vector<int>& lev, std::vector<int>& nodes, std::vector<int>& lvl_ptr, std::vector<int>& lvl_off, int& maxlev,
                   const std::vector<long long>& len_ptr, std::vector<int>& lvl_g) {
    nodes.resize((size_t)n1);
    std::iota(nodes.begin(), nodes.end(), 0);
    std::stable_sort(nodes.begin(), nodes.end(), [&](int a, int b) {
      if (tree_of[a] != tree_of[b]) return tree_of[a] < tree_of[b];
      return lev[a] < lev[b];
    });
    lvl_ptr.assign((size_t)ntrees + 1, 0);
    lvl_off.clear();
    int cur_tree = -1, cur_lev = -1;
    for (int idx = 0; idx < n1; ++idx) {
      const int a = nodes[idx];
      if (tree_of[a] != cur_tree) {
        if (cur_tree >= 0) lvl_off.push_back(idx);         // sentinel of the previous tree
        cur_tree = tree_of[a]; cur_lev = -1;
        lvl_ptr[cur_tree] = (int)lvl_off.size();
      }
      if (lev[a] != cur_lev) { lvl_off.push_back(idx); cur_lev = lev[a]; maxlev = std::max(maxlev, cur_lev + 1); }
    }
    lvl_off.push_back(n1);
    lvl_ptr[ntrees] = (int)lvl_off.size();
    // lanes per row of a level: the power of two next to the level's mean number of nonzeros per node (1 .. 64)
    lvl_g.assign(lvl_off.size(), 1);
    for (size_t l = 0; l + 1 < lvl_off.size(); ++l) {
      const int cnt = lvl_off[l + 1] - lvl_off[l];
      if (cnt <= 0) continue;
      long long tot = 0;
      for (int idx = lvl_off[l]; idx < lvl_off[l + 1]; ++idx) tot += len_ptr[(size_t)nodes[idx] + 1] - len_ptr[nodes[idx]];
      const double mean = (double)tot / cnt;
      int g = 1;
      while (g < 64 && g < mean) g <<= 1;
      lvl_g[l] = g;
    }
  };
  std::vector<int> nf, nb, lpf, lpb, lof, lob, lgf, lgb;
  int mlf = 0, mlb = 0;
  // per-node entry counts of the two sweeps: forward = row of L11, backward = leading rows of the column
  std::vector<long long> cnt_b((size_t)n1 + 1, 0);
  for (int j = 0; j < n1; ++j) {
    long long c = 0;
    for (long long p = Lp[j]; p < Lp[j + 1]; ++p) c += Li[p] < n1;
    cnt_b[(size_t)j + 1] = cnt_b[j] + c;
  }
  order(lev_f, nf, lpf, lof, mlf, r11, lgf);
  order(lev_b, nb, lpb, lob, mlb, cnt_b, lgb);
  max_levels = std::max(mlf, mlb);
  // local index of a node = its position inside its tree in the sweep's processing order
  std::vector<int> tree_first((size_t)ntrees, 0);
  max_nodes = 0;
  {
    std::vector<int> cnt((size_t)ntrees, 0);
    for (int j = 0; j < n1; ++j) cnt[tree_of[j]]++;
    int acc = 0;
    for (int t = 0; t < ntrees; ++t) { tree_first[t] = acc; acc += cnt[t]; max_nodes = std::max(max_nodes, cnt[t]); }
  }
  std::vector<int> pos_f((size_t)n1), pos_b((size_t)n1);
  for (int idx = 0; idx < n1; ++idx) { pos_f[nf[idx]] = idx - tree_first[tree_of[nf[idx]]]; pos_b[nb[idx]] = idx - tree_first[tree_of[nb[idx]]]; }
  // forward stream: row of L11 of the node in slot idx, local column indices
  std::vector<long long> fp((size_t)n1 + 1, 0), bp((size_t)n1 + 1, 0), tp((size_t)n1 + 1, 0);
  std::vector<int> fc((size_t)r11[n1]), bc((size_t)cnt_b[n1]), tr((size_t)r21[k]);
  std::vector<double> fv((size_t)r11[n1]), bv((size_t)cnt_b[n1]), tv((size_t)r21[k]);
  for (int idx = 0; idx < n1; ++idx) {
    const int i = nf[idx];
    long long q = fp[idx];
    for (long long r = r11[i]; r < r11[i + 1]; ++r, ++q) { fc[(size_t)q] = pos_f[c11[(size_t)r]]; fv[(size_t)q] = w11[(size_t)r]; }
    fp[(size_t)idx + 1] = q;
  }
  // backward stream: leading rows of the column of the node in slot idx, local row indices; tail rows by column for w
  for (int idx = 0; idx < n1; ++idx) {
    const int j = nb[idx];
    long long q = bp[idx];
    for (long long p = Lp[j]; p < Lp[j + 1]; ++p)
      if (Li[p] < n1) { bc[(size_t)q] = pos_b[Li[p]]; bv[(size_t)q] = Lx[p]; ++q; }
    bp[(size_t)idx + 1] = q;
  }
  std::vector<int> long_cols;
  for (int j = 0; j < n1; ++j) {
    long long q = tp[j];
    for (long long p = Lp[j]; p < Lp[j + 1]; ++p)
      if (Li[p] >= n1) { tr[(size_t)q] = Li[p] - n1; tv[(size_t)q] = Lx[p]; ++q; }
    tp[(size_t)j + 1] = q;
    if (q - tp[j] > kLongColumn) long_cols.push_back(j);
  }
  n_long = (int)long_cols.size();
  if (debug) {
    long long mx_f = 0, mx_b = 0;
    std::vector<int> cnt((size_t)ntrees, 0);
    for (int j = 0; j < n1; ++j) cnt[tree_of[j]]++;
    int acc = 0;
    for (int t = 0; t < ntrees; ++t) { mx_f = std::max(mx_f, fp[(size_t)acc + cnt[t]] - fp[acc]); mx_b = std::max(mx_b, bp[(size_t)acc + cnt[t]] - bp[acc]); acc += cnt[t]; }
    fprintf(stderr, "[lead debug] n1 %d k %d trees %d max_nodes %d max_levels %d nnz11 %lld (fwd) %lld (bwd) max per tree %lld / %lld nnz21 %lld\n", n1, k, ntrees,
            max_nodes, max_levels, (long long)fp[n1], (long long)bp[n1], mx_f, mx_b, (long long)r21[k]);
  }
  // cost model: per level two dependent global-memory latencies (~2 us) in the deepest tree, per sweep, plus the streaming part
  est_us = 2.0 * 2.0 * max_levels + 40.0 + (double)nnz * 2e-4;
  nnz21 = r21[k];
  if (max_nodes > 6144 || max_levels > 2048) {                                        // LDS budget of one wavefront's tree
    est_us = 1e30;
    if (allow_hybrid && l21_pays()) {
      int rc_;
      if ((rc_ = to_device(rp21, r21)) || (rc_ = to_device(ci21, c21)) || (rc_ = to_device(v21, w21)) ||
          (rc_ = to_device(tptr, tp)) || (rc_ = to_device(tri, tr)) || (rc_ = to_device(tv_, tv)) || (rc_ = to_device(long_cols_d, long_cols)) ||
          (rc_ = hybrid_buffers(*this))) { release(); return rc_; }
      hybrid = true;
    }
    return CUADMM_OK;
  }
  lds_bytes = sizeof(double) * (size_t)max_nodes + sizeof(int) * (2 * (size_t)max_levels + 2);
  // classes by the LDS a tree needs with its stream resident (the larger of the two sweeps): small trees share a CU in numbers,
  // the few big ones get a launch of their own, anything beyond one workgroup's LDS keeps the streaming kernels
  std::vector<int> t_stream;
  std::vector<LeadTreeDesc> dsf, dsb, dbf, dbb;        // small / big trees, forward / backward sweep
  std::vector<int> mfirst, mcnt;                       // micro trees (lead_solve.h)
  bool split_micro = false;
  lds_small = lds_big = 0;
  // (measured and dropped: a launch of their own for the trees that need <= 4 KB -- eight workgroups per CU instead of two.  The two launches
  // serialise, and the big trees decide the second one: PlanarHand_N=10 116 -> 2 x 50 us per sweep, PushBox N = 30 / 50 +10 us per sweep.)
  {
    std::vector<int> cnt((size_t)ntrees, 0);
    for (int j = 0; j < n1; ++j) cnt[tree_of[j]]++;
    std::vector<size_t> need_of((size_t)ntrees);
    {
      int acc = 0;
      for (int t = 0; t < ntrees; ++t) {
        const long long nzf = fp[(size_t)acc + cnt[t]] - fp[acc], nzb = bp[(size_t)acc + cnt[t]] - bp[acc];
        const int nlf = lpf[(size_t)t + 1] - lpf[t] - 1, nlb = lpb[(size_t)t + 1] - lpb[t] - 1;
        const long long nz = std::max(nzf, nzb);
        const int nlev_t = std::max(nlf, nlb);
        need_of[t] = sizeof(double) * (2 * (size_t)cnt[t] + (size_t)nz) + sizeof(int) * (2 * (size_t)cnt[t] + 1 + (size_t)nz + 2 * (size_t)nlev_t + 2) + 16;
        acc += cnt[t];
      }
    }
    // The bound between the two resident classes.  One launch serves both, so its LDS per workgroup is the larger of the biggest tree and of
    // four times the largest SMALL one: a low bound lets more workgroups share a CU but turns small trees into workgroups of their own.  Taken
    // from {4, 8, 16} KB by the rounds of workgroups the launch needs (workgroups / (CUs x workgroups per CU by LDS)) -- the quantity that
    // ordered every measurement (profiles/r05_lead_small_kb.log: PushBox N = 30 3.1 rounds at 16 KB, 2.6 at 4 KB: 0.93 -> 0.83 ms per
    // iteration; N = 50 4.8 against 5.8 and PlanarHand_N=10 11.0 against 12.7: 16 KB stays).  The arithmetic of a tree does not depend on
    // its class: the same bits either way.
    size_t bound = (size_t)small_kb * 1024;
    if (small_kb <= 0) {
      double best = 1e300;
      for (const size_t cand : {(size_t)16 * 1024, (size_t)8 * 1024, (size_t)4 * 1024}) {          // ties: the larger bound
        size_t ls = 0, lb = 0;
        long long ns = 0, nb = 0;
        for (int t = 0; t < ntrees; ++t) {
          if (stream_only || need_of[t] > kMaxLdsBytes - 1024) continue;
          if (need_of[t] <= cand) { ++ns; ls = std::max(ls, need_of[t]); } else { ++nb; lb = std::max(lb, need_of[t]); }
        }
        const size_t lds = std::max<size_t>(std::max(lb, 4 * ((ls + 7) / 8 * 8)), 1024);
        const double per_cu = (double)std::min<size_t>(8, std::max<size_t>(1, kMaxLdsBytes / lds));
        const double rounds = ((double)nb + (double)((ns + 3) / 4)) / (256.0 * per_cu);
        if (rounds < 0.98 * best) { best = rounds; bound = cand; }
      }
    }
    // a launch of their own pays from a few thousand micro trees on (below that they are a fraction of one round of workgroups)
    {
      int nm = 0;
      for (int t = 0; t < ntrees; ++t) nm += cnt[t] <= 2;
      split_micro = !stream_only && nm >= 8192;
    }
    int acc = 0;
    for (int t = 0; t < ntrees; ++t) {
      const long long nzf = fp[(size_t)acc + cnt[t]] - fp[acc], nzb = bp[(size_t)acc + cnt[t]] - bp[acc];
      const int nlf = lpf[(size_t)t + 1] - lpf[t] - 1, nlb = lpb[(size_t)t + 1] - lpb[t] - 1;
      const size_t need = need_of[t];
      const LeadTreeDesc df{lpf[t], nlf, acc, cnt[t], (int)nzf, 0, fp[acc]}, db{lpb[t], nlb, acc, cnt[t], (int)nzb, 0, bp[acc]};
      if (stream_only || need > kMaxLdsBytes - 1024) t_stream.push_back(t);
      else if (split_micro && cnt[t] <= 2) { mfirst.push_back(acc); mcnt.push_back(cnt[t]); }
      else if (need <= bound) { dsf.push_back(df); dsb.push_back(db); lds_small = std::max(lds_small, need); }
      else { dbf.push_back(df); dbb.push_back(db); lds_big = std::max(lds_big, need); }
      acc += cnt[t];
    }
  }
  n_small = (int)dsf.size(); n_big = (int)dbf.size(); n_stream = (int)t_stream.size(); n_micro = (int)mfirst.size();
  if (debug) fprintf(stderr, "[lead debug] resident trees: %d small (%zu B), %d big (%zu B), %d streaming, %d micro\n", n_small, lds_small, n_big, lds_big, n_stream, n_micro);
  int rc;
  if ((rc = to_device(micro_first, mfirst)) || (rc = to_device(micro_cnt, mcnt))) return rc;
  {
    LeadTreeDesc *a0 = nullptr, *a1 = nullptr, *a2 = nullptr, *a3 = nullptr;
    if ((rc = to_device(a0, dsf)) || (rc = to_device(a1, dsb)) || (rc = to_device(a2, dbf)) || (rc = to_device(a3, dbb)) || (rc = to_device(trees_stream, t_stream))) return rc;
    desc_small_f = a0; desc_small_b = a1; desc_big_f = a2; desc_big_b = a3;
  }
  // the attribute is per kernel and process-wide: always the hardware maximum, so that a second solver with smaller trees cannot lower it
  if (lds_big > 48 * 1024) {
    CUADMM_HIP_TRY(allow_max_dynamic_lds(reinterpret_cast<const void*>(lead_sweep_lds_kernel<false, 4>)));
    CUADMM_HIP_TRY(allow_max_dynamic_lds(reinterpret_cast<const void*>(lead_sweep_lds_kernel<true, 4>)));
  }
  if (std::max(lds_big, 4 * ((lds_small + 7) / 8 * 8)) > 48 * 1024) {
    CUADMM_HIP_TRY(allow_max_dynamic_lds(reinterpret_cast<const void*>(lead_sweep_merged_kernel<false>)));
    CUADMM_HIP_TRY(allow_max_dynamic_lds(reinterpret_cast<const void*>(lead_sweep_merged_kernel<true>)));
  }
  // cost model of the resident variant: two bulk round trips, then ~0.15 us of LDS work per level
  if (n_stream == 0) est_us = 2.0 * (6.0 + 0.15 * max_levels) * (n_big > 0 ? 2.0 : 1.0) + 40.0 + (double)nnz * 2e-4;
  if ((rc = to_device(rp21, r21)) || (rc = to_device(ci21, c21)) || (rc = to_device(v21, w21)) ||
      (rc = to_device(fptr, fp)) || (rc = to_device(fci, fc)) || (rc = to_device(fv_, fv)) ||
      (rc = to_device(bptr, bp)) || (rc = to_device(bci, bc)) || (rc = to_device(bv_, bv)) ||
      (rc = to_device(tptr, tp)) || (rc = to_device(tri, tr)) || (rc = to_device(tv_, tv)) || (rc = to_device(long_cols_d, long_cols)) ||
      (rc = to_device(D1, pinv_pivots(D, n1, pinv_tol))) || (rc = to_device(nodes_f, nf)) || (rc = to_device(nodes_b, nb)) ||
      (rc = to_device(lvl_ptr_f, lpf)) || (rc = to_device(lvl_ptr_b, lpb)) || (rc = to_device(lvl_off_f, lof)) || (rc = to_device(lvl_off_b, lob)) ||
      (rc = to_device(lvl_g_f, lgf)) || (rc = to_device(lvl_g_b, lgb)))
    return rc;
  CUADMM_HIP_TRY(hipMalloc(&wvec, sizeof(double) * (size_t)n1));
  if (lds_bytes > 48 * 1024) {
    CUADMM_HIP_TRY(allow_max_dynamic_lds(reinterpret_cast<const void*>(lead_forward_kernel)));
    CUADMM_HIP_TRY(allow_max_dynamic_lds(reinterpret_cast<const void*>(lead_backward_kernel)));
  }
  // (round 5: big and small trees share one launch, lead_sweep_merged_kernel: no side stream, no fork / join events)
  ready = true;
  return CUADMM_OK;
}

// Dense tree tops (lead_solve.h).  T = leading nodes of height >= `level` in the elimination forest (height of a node = its level in the
// forward sweep): upward closed, one connected top per tree.  The factor is taken in the order B | T (tree after tree) | K; build_core then
// sees B as the leading block and T | K as its tail rows, and the middle stage is built here: W_t = L_tt^-1 per tree (rows of the unit
// lower-triangular inverse from the sparse rows of L_tt: W[r] = e_r - sum_c L[r][c] W[c]), packed by rows and, transposed, by columns;
// L_KT by rows and by columns.  Every inverse is checked (L_tt (W_t v) = v to 1e-9) before it is trusted: CUADMM_ERR_FACTOR otherwise.
int LeadSolve::build_tops(int m_, int k_, const int64_t* Lp, const int* Li, const double* Lx, const double* D, int level) {
  const int n1o = m_ - k_;
  std::vector<int> parent((size_t)n1o, -1), height((size_t)n1o, 0), root((size_t)n1o);
  for (int j = 0; j < n1o; ++j) {
    int pmin = -1;
    for (long long p = Lp[j]; p < Lp[j + 1]; ++p) if (Li[p] < n1o && (pmin < 0 || Li[p] < pmin)) pmin = Li[p];
    parent[j] = pmin;
    if (pmin >= 0) height[pmin] = std::max(height[pmin], height[j] + 1);
  }
  for (int j = n1o - 1; j >= 0; --j) root[j] = parent[j] < 0 ? j : root[parent[j]];
  // T in the order (tree, index): the nodes of one top are contiguous
  std::vector<int> tnodes;
  for (int j = 0; j < n1o; ++j) if (height[j] >= level) tnodes.push_back(j);
  const int nt = (int)tnodes.size();
  if (nt == 0 || nt == n1o) return CUADMM_ERR_FACTOR;                 // nothing to cut / nothing left: not this path
  std::stable_sort(tnodes.begin(), tnodes.end(), [&](int a, int b) { return root[a] < root[b]; });
  const int nB = n1o - nt;
  std::vector<int> newid((size_t)n1o), ridh((size_t)m_);
  {
    int pb = 0;
    for (int j = 0; j < n1o; ++j) if (height[j] < level) { newid[j] = pb; ridh[pb++] = j; }
    for (int t = 0; t < nt; ++t) { newid[tnodes[t]] = nB + t; ridh[(size_t)nB + t] = tnodes[t]; }
    for (int r = n1o; r < m_; ++r) ridh[r] = r;
  }
  // blocks of T
  std::vector<int> blk_ptr{0};
  for (int t = 1; t <= nt; ++t) if (t == nt || root[tnodes[t]] != root[tnodes[t - 1]]) blk_ptr.push_back(t);
  const int nblk = (int)blk_ptr.size() - 1;
  long long tri_total = 0;
  int bmax = 0;
  for (int q = 0; q < nblk; ++q) { const long long n = blk_ptr[q + 1] - blk_ptr[q]; tri_total += n * (n + 1) / 2; bmax = std::max(bmax, (int)n); }
  if (tri_total * 16 > kLeadTopsMaxBytes) return CUADMM_ERR_FACTOR;     // both triangles beyond the planner's own bound (common.h): not this path
  // the B columns in the new order (rows: B ascending, then T, then K -- T rows sorted by their new index)
  std::vector<int64_t> Lp2((size_t)nB + 1, 0);
  std::vector<int> Li2;
  std::vector<double> Lx2, D2((size_t)m_);
  for (int p = 0; p < m_; ++p) D2[p] = D[ridh[p]];
  {
    long long nz = 0;
    for (int pb = 0; pb < nB; ++pb) { const int j = ridh[pb]; nz += Lp[j + 1] - Lp[j]; }
    Li2.resize((size_t)nz); Lx2.resize((size_t)nz);
    std::vector<std::pair<int, double>> trow;
    long long q = 0;
    for (int pb = 0; pb < nB; ++pb) {
      const int j = ridh[pb];
      trow.clear();
      for (long long p = Lp[j]; p < Lp[j + 1]; ++p) {
        const int i = Li[p];
        if (i < n1o && height[i] < level) { Li2[(size_t)q] = newid[i]; Lx2[(size_t)q] = Lx[p]; ++q; }
        else if (i < n1o) trow.emplace_back(newid[i], Lx[p]);
      }
      // B rows ascending in the old order are ascending in the new one; T rows are sorted here
      std::sort(trow.begin(), trow.end(), [](const std::pair<int, double>& a, const std::pair<int, double>& b) { return a.first < b.first; });
      for (const auto& e : trow) { Li2[(size_t)q] = e.first; Lx2[(size_t)q] = e.second; ++q; }
      for (long long p = Lp[j]; p < Lp[j + 1]; ++p) if (Li[p] >= n1o) { Li2[(size_t)q] = Li[p]; Lx2[(size_t)q] = Lx[p]; ++q; }
      Lp2[(size_t)pb + 1] = q;
    }
  }
  int rc = build_core(m_, nt + k_, Lp2.data(), Li2.data(), Lx2.data(), D2.data(), false);
  if (rc) return rc;
  if (!ready) return CUADMM_ERR_FACTOR;                                // the rest is still too deep / too big for the sweeps
  ready = false;
  std::vector<int64_t>().swap(Lp2); std::vector<int>().swap(Li2); std::vector<double>().swap(Lx2);
  // ---- the middle stage
  // sparse rows of L_TT (T-local indices) and L_KT from the T columns
  std::vector<long long> rpt((size_t)nt + 1, 0), krp((size_t)k_ + 1, 0), kcp((size_t)nt + 1, 0);
  for (int t = 0; t < nt; ++t) {
    const int j = tnodes[t];
    for (long long p = Lp[j]; p < Lp[j + 1]; ++p) {
      const int i = Li[p];
      if (i < n1o) rpt[(size_t)(newid[i] - nB) + 1]++;
      else { krp[(size_t)(i - n1o) + 1]++; kcp[(size_t)t + 1]++; }
    }
  }
  for (int t = 0; t < nt; ++t) { rpt[(size_t)t + 1] += rpt[t]; kcp[(size_t)t + 1] += kcp[t]; }
  for (int i = 0; i < k_; ++i) krp[(size_t)i + 1] += krp[i];
  std::vector<int> cit((size_t)rpt[nt]), kci((size_t)krp[k_]), kri((size_t)kcp[nt]);
  std::vector<double> vt((size_t)rpt[nt]), kv((size_t)krp[k_]), kcv((size_t)kcp[nt]);
  {
    std::vector<long long> ft(rpt.begin(), rpt.end() - 1), fk(krp.begin(), krp.end() - 1);
    for (int t = 0; t < nt; ++t) {                      // columns ascending => column indices ascending inside every row
      const int j = tnodes[t];
      long long qc = kcp[t];
      for (long long p = Lp[j]; p < Lp[j + 1]; ++p) {
        const int i = Li[p];
        if (i < n1o) { const long long q = ft[(size_t)(newid[i] - nB)]++; cit[(size_t)q] = t; vt[(size_t)q] = Lx[p]; }
        else {
          const long long q = fk[(size_t)(i - n1o)]++;
          kci[(size_t)q] = t; kv[(size_t)q] = Lx[p];
          kri[(size_t)qc] = i - n1o; kcv[(size_t)qc] = Lx[p]; ++qc;
        }
      }
    }
  }
  // packed inverses, block after block, on the host pool
  std::vector<long long> offf((size_t)nt), offb((size_t)nt), base((size_t)nblk + 1, 0);
  std::vector<int> begf((size_t)nt), lenb((size_t)nt);
  for (int q = 0; q < nblk; ++q) { const long long n = blk_ptr[q + 1] - blk_ptr[q]; base[(size_t)q + 1] = base[q] + n * (n + 1) / 2; }
  for (int q = 0; q < nblk; ++q) {
    const int t0 = blk_ptr[q], n = blk_ptr[q + 1] - t0;
    for (int r = 0; r < n; ++r) {
      offf[(size_t)t0 + r] = base[q] + (long long)r * (r + 1) / 2;
      begf[(size_t)t0 + r] = t0;
      offb[(size_t)t0 + r] = base[q] + (long long)r * n - (long long)r * (r - 1) / 2;     // rows of W^T: n, n - 1, ... entries
      lenb[(size_t)t0 + r] = n - r;
    }
  }
  std::vector<double> wf((size_t)tri_total), wb((size_t)tri_total);
  std::atomic<int> next{0};
  std::atomic<int> bad{0};
  auto work = [&]() {
    for (;;) {
      const int q = next.fetch_add(1);
      if (q >= nblk) return;
      try {
      const int t0 = blk_ptr[q], n = blk_ptr[q + 1] - t0;
      double* W = wf.data() + base[q];
      for (int r = 0; r < n; ++r) {
        double* wr = W + (long long)r * (r + 1) / 2;
        std::fill(wr, wr + r, 0.0);
        wr[r] = 1.0;
        for (long long e = rpt[(size_t)t0 + r]; e < rpt[(size_t)t0 + r + 1]; ++e) {
          const int c = cit[(size_t)e] - t0;            // a column of the same block, c < r
          const double l = vt[(size_t)e];
          const double* wc = W + (long long)c * (c + 1) / 2;
          for (int x = 0; x <= c; ++x) wr[x] -= l * wc[x];
        }
      }
      // the transposed copy: row r of W^T = column r of W, from the diagonal down
      double* WT = wb.data() + base[q];
      for (int r = 0; r < n; ++r) {
        const double* wr = W + (long long)r * (r + 1) / 2;
        for (int c = 0; c <= r; ++c) WT[(long long)c * n - (long long)c * (c - 1) / 2 + (r - c)] = wr[c];
      }
      // check: L_tt (W v) = v
      std::vector<double> v((size_t)n), tvec((size_t)n);
      unsigned long long seed = 0x9e3779b97f4a7c15ull + (unsigned long long)q;
      for (int r = 0; r < n; ++r) { seed = seed * 6364136223846793005ull + 1442695040888963407ull; v[r] = 0.5 + (double)(seed >> 11) * (1.0 / 9007199254740992.0); }
      for (int r = 0; r < n; ++r) {
        const double* wr = W + (long long)r * (r + 1) / 2;
        double acc = 0.0;
        for (int c = 0; c <= r; ++c) acc += wr[c] * v[c];
        tvec[r] = acc;
      }
      double err = 0.0;
      for (int r = 0; r < n; ++r) {
        double acc = tvec[r];
        for (long long e = rpt[(size_t)t0 + r]; e < rpt[(size_t)t0 + r + 1]; ++e) acc += vt[(size_t)e] * tvec[cit[(size_t)e] - t0];
        err = std::max(err, std::fabs(acc - v[r]));
      }
      if (!(err <= 1e-9)) bad.fetch_add(1);
      } catch (const std::bad_alloc&) { bad.fetch_add(1); }      // (a thread must not let it escape)
    }
  };
  {
    // on the SHARED host pool (no threads of its own: the ranks of an in-process group build their tops at the same time, and a
    // std::thread that cannot start would unwind through joinable ones); blocks are handed out by the counter, so a pool that is busy
    // -- its caller then runs the chunks inline -- changes nothing but the time
    const int nth = std::max(1, std::min(cuadmm_host_pool_threads(), nblk));
    auto* wp = &work;
    cuadmm_host_parallel_for(nth, [](int, void* p) { (*static_cast<decltype(wp)>(p))(); }, wp);
  }
  if (bad.load() > 0) { set_error("lead_solve: %d of %d dense tree tops fail the check of their inverse", bad.load(), nblk); return CUADMM_ERR_FACTOR; }
  std::vector<double> dt((size_t)nt);
  for (int t = 0; t < nt; ++t) dt[t] = D[tnodes[t]];
  dt = pinv_pivots(dt.data(), nt, pinv_tol);
  // L_TT by columns for the refinement of the backward solve (the rows of L_TT^T)
  std::vector<long long> cpt((size_t)nt + 1, 0);
  for (long long e = 0; e < rpt[nt]; ++e) cpt[(size_t)cit[(size_t)e] + 1]++;
  for (int t = 0; t < nt; ++t) cpt[(size_t)t + 1] += cpt[t];
  std::vector<int> rit((size_t)rpt[nt]);
  std::vector<double> vct((size_t)rpt[nt]);
  {
    std::vector<long long> fc(cpt.begin(), cpt.end() - 1);
    for (int r = 0; r < nt; ++r)
      for (long long e = rpt[r]; e < rpt[(size_t)r + 1]; ++e) { const long long q = fc[(size_t)cit[(size_t)e]]++; rit[(size_t)q] = r; vct[(size_t)q] = vt[(size_t)e]; }
  }
  if ((rc = to_device(tt_rp, rpt)) || (rc = to_device(tt_ci, cit)) || (rc = to_device(tt_v, vt)) || (rc = to_device(tt_cp, cpt)) || (rc = to_device(tt_ri, rit)) ||
      (rc = to_device(tt_cv, vct)))
    return rc;
  if ((rc = to_device(rid, ridh)) || (rc = to_device(Wf, wf)) || (rc = to_device(Wb, wb)) || (rc = to_device(wf_off, offf)) || (rc = to_device(wb_off, offb)) ||
      (rc = to_device(wf_beg, begf)) || (rc = to_device(wb_len, lenb)) || (rc = to_device(kt_rp, krp)) || (rc = to_device(kt_ci, kci)) || (rc = to_device(kt_v, kv)) ||
      (rc = to_device(tk_cp, kcp)) || (rc = to_device(tk_ri, kri)) || (rc = to_device(tk_v, kcv)) || (rc = to_device(DT, dt)))
    return rc;
  CUADMM_HIP_TRY(hipMalloc(&xp, sizeof(double) * (size_t)m_));
  CUADMM_HIP_TRY(hipMalloc(&zext, sizeof(double) * (size_t)(nt + k_)));
  CUADMM_HIP_TRY(hipMalloc(&xext, sizeof(double) * (size_t)(nt + k_)));
  CUADMM_HIP_TRY(hipMalloc(&zT, sizeof(double) * (size_t)nt));
  CUADMM_HIP_TRY(hipMalloc(&uT, sizeof(double) * (size_t)nt));
  nT = nt; k_tail = k_; tops = true;
  tops_bytes = tri_total * 16; tops_blocks = nblk; tops_max = bmax;
  // cost model: the sweeps of the shallow rest (build_core's), the SpMVs (12 B per nonzero at 3.5 TB/s), the two dense passes at 4.5 TB/s, launches
  est_us += 2.0 * (double)(krp[k_]) * 12.0 / 3.5e6 + (double)tops_bytes / 4.5e6 + 50.0;
  if (debug)
    fprintf(stderr, "[lead debug] dense tree tops at height %d: %d nodes in %d blocks (largest %d), %.0f MB of inverses; L_KT %lld nonzeros; rest: %d nodes, depth <= %d\n",
            level, nt, nblk, bmax, (double)tops_bytes / 1e6, (long long)krp[k_], nB, max_levels);
  ready = true;
  return CUADMM_OK;
}

// the sweeps over the leading forest, one direction: big trees (four wavefronts each) and small ones (one wavefront each, four per
// workgroup) in ONE launch when both exist, the streaming kernels for trees beyond a workgroup's LDS
static void launch_sweeps(const LeadSolve& L, bool backward, const double* ax, const double* asmc, const double* b, double isig, double* x, hipStream_t st,
                          double* yout = nullptr) {
  double* const ynul = nullptr;
  const bool merged = L.n_big > 0 && L.n_small > 0;
  const int small_doubles = (int)((L.lds_small + 7) / 8);
  const size_t lds_merged = std::max(L.lds_big, 4 * sizeof(double) * (size_t)small_doubles);
  const unsigned grid_merged = (unsigned)(L.n_big + (L.n_small + 3) / 4);
  const double* nul = nullptr;
  if (L.n_micro > 0) {
    if (!backward)
      hipLaunchKernelGGL(lead_micro_kernel<false>, dim3((unsigned)((L.n_micro + 255) / 256)), dim3(256), 0, st, L.n_micro, L.micro_first, L.micro_cnt, L.nodes_f, L.fptr,
                         L.fci, L.fv_, ax, asmc, b, isig, nul, nul, x, L.rid, ynul);
    else
      hipLaunchKernelGGL(lead_micro_kernel<true>, dim3((unsigned)((L.n_micro + 255) / 256)), dim3(256), 0, st, L.n_micro, L.micro_first, L.micro_cnt, L.nodes_b, L.bptr,
                         L.bci, L.bv_, nul, nul, nul, 0.0, L.D1, L.wvec, x, L.rid, yout);
  }
  if (!backward) {
    if (merged)
      hipLaunchKernelGGL(lead_sweep_merged_kernel<false>, dim3(grid_merged), dim3(256), lds_merged, st, static_cast<const LeadTreeDesc*>(L.desc_big_f), L.n_big,
                         static_cast<const LeadTreeDesc*>(L.desc_small_f), L.n_small, small_doubles, L.lvl_off_f, L.lvl_g_f, L.nodes_f, L.fptr, L.fci, L.fv_, ax, asmc, b, isig,
                         nul, nul, x, L.rid, ynul);
    else if (L.n_small > 0)
      hipLaunchKernelGGL((lead_sweep_lds_kernel<false, 1>), dim3(L.n_small), dim3(64), L.lds_small, st, static_cast<const LeadTreeDesc*>(L.desc_small_f), L.lvl_off_f,
                         L.lvl_g_f, L.nodes_f, L.fptr, L.fci, L.fv_, ax, asmc, b, isig, nul, nul, x, L.rid, ynul);
    else if (L.n_big > 0)
      hipLaunchKernelGGL((lead_sweep_lds_kernel<false, 4>), dim3(L.n_big), dim3(256), L.lds_big, st, static_cast<const LeadTreeDesc*>(L.desc_big_f), L.lvl_off_f, L.lvl_g_f,
                         L.nodes_f, L.fptr, L.fci, L.fv_, ax, asmc, b, isig, nul, nul, x, L.rid, ynul);
    if (L.n_stream > 0) hipLaunchKernelGGL(lead_forward_kernel, dim3(L.n_stream), dim3(64), L.lds_bytes, st, L.lvl_ptr_f, L.lvl_off_f, L.lvl_g_f, L.nodes_f, L.fptr, L.fci, L.fv_,
                                           ax, asmc, b, isig, x, L.max_nodes, L.max_levels, L.trees_stream, L.rid);
  } else {
    if (merged)
      hipLaunchKernelGGL(lead_sweep_merged_kernel<true>, dim3(grid_merged), dim3(256), lds_merged, st, static_cast<const LeadTreeDesc*>(L.desc_big_b), L.n_big,
                         static_cast<const LeadTreeDesc*>(L.desc_small_b), L.n_small, small_doubles, L.lvl_off_b, L.lvl_g_b, L.nodes_b, L.bptr, L.bci, L.bv_,
                         nul, nul, nul, 0.0, L.D1, L.wvec, x, L.rid, yout);
    else if (L.n_small > 0)
      hipLaunchKernelGGL((lead_sweep_lds_kernel<true, 1>), dim3(L.n_small), dim3(64), L.lds_small, st, static_cast<const LeadTreeDesc*>(L.desc_small_b), L.lvl_off_b,
                         L.lvl_g_b, L.nodes_b, L.bptr, L.bci, L.bv_, nul, nul, nul, 0.0, L.D1, L.wvec, x, L.rid, yout);
    else if (L.n_big > 0)
      hipLaunchKernelGGL((lead_sweep_lds_kernel<true, 4>), dim3(L.n_big), dim3(256), L.lds_big, st, static_cast<const LeadTreeDesc*>(L.desc_big_b), L.lvl_off_b, L.lvl_g_b,
                         L.nodes_b, L.bptr, L.bci, L.bv_, nul, nul, nul, 0.0, L.D1, L.wvec, x, L.rid, yout);
    if (L.n_stream > 0) hipLaunchKernelGGL(lead_backward_kernel, dim3(L.n_stream), dim3(64), L.lds_bytes, st, L.lvl_ptr_b, L.lvl_off_b, L.lvl_g_b, L.nodes_b, L.bptr, L.bci, L.bv_,
                                           L.D1, L.wvec, x, L.max_nodes, L.max_levels, L.trees_stream, L.rid, yout);
  }
}

int LeadSolve::solve(const double* ax, const double* asmc, const double* b, double isig, double* y, TailSolve& tail, hipStream_t st) const {
  if (!ready) { set_error("lead_solve: not built"); return CUADMM_ERR_INVALID; }
  if (tops) return solve_tops(ax, asmc, b, isig, y, tail, st);
  launch_sweeps(*this, false, ax, asmc, b, isig, y, st);
  hipLaunchKernelGGL(lead_tail_rhs_kernel, dim3((k * 8 + 255) / 256), dim3(256), 0, st, k, n1, rp21, ci21, v21, ax, asmc, b, isig, y, tail.vin);
  CUADMM_HIP_TRY(hipGetLastError());
  int rc = tail.solve_device(st);                    // vin <- L22^-T D2^-1 L22^-1 vin (padding beyond k stays zero)
  if (rc) return rc;
  const bool copied = launch_l21t(n1, tptr, tri, tv_, tail.vin, wvec, n_long, long_cols_d, st, y + n1, k);     // ... and the solved tail into y
  launch_sweeps(*this, true, ax, asmc, b, isig, y, st);
  CUADMM_HIP_TRY(hipGetLastError());
  // (never hipMemcpyAsync: the runtime's device-to-device copy is a blit behind ~15 us of command-processor work -- kernel trace of
  // pendulum N = 80, round 5)
  if (!copied) hipLaunchKernelGGL(lead_copy_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, st, tail.vin, y + n1, k);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

// Dense tree tops (lead_solve.h): B sweeps, the block-diagonal middle stage, the tail, and back.
int LeadSolve::solve_tops(const double* ax, const double* asmc, const double* b, double isig, double* y, TailSolve& tail, hipStream_t st) const {
  const int kt = k_tail, kext = k;                    // k = |T| + tail: what the B sweeps see as "the tail rows"
  launch_sweeps(*this, false, ax, asmc, b, isig, xp, st);                                                       // z_B
  hipLaunchKernelGGL(lead_ext_rhs_kernel<32>, dim3((unsigned)(((long long)kext * 32 + 255) / 256)), dim3(256), 0, st, kext, n1, rp21, ci21, v21, ax, asmc, b, isig, rid,
                     xp, zext);                                                                                  // r_T - L_TB z_B | r_K - L_KB z_B
  hipLaunchKernelGGL(tops_gemv_kernel, dim3((unsigned)((nT + 3) / 4)), dim3(256), 0, st, nT, wf_off, wf_beg, (const int*)nullptr, Wf, zext, zT, 0);   // z_T = W_T (...)
  if (tops_refine) {                                                                                             // z_T += W_T (r - L_TT z_T)
    hipLaunchKernelGGL(tops_resid_kernel, dim3((unsigned)(((long long)nT * 32 + 255) / 256)), dim3(256), 0, st, nT, tt_rp, tt_ci, tt_v, zT, zext, uT);
    hipLaunchKernelGGL(tops_gemv_kernel, dim3((unsigned)((nT + 3) / 4)), dim3(256), 0, st, nT, wf_off, wf_beg, (const int*)nullptr, Wf, uT, zT, 1);
  }
  const int* pz = tail.z_scatter();                   // the tail reads z in its factor's pivoting order: written there, not gathered
  hipLaunchKernelGGL(tops_k_rhs_kernel, dim3((unsigned)((kt + 3) / 4)), dim3(256), 0, st, kt, kt_rp, kt_ci, kt_v, zT, zext + nT, tail.vin, pz);   // z_K -= L_KT z_T
  CUADMM_HIP_TRY(hipGetLastError());
  tail.vin_pivot = pz != nullptr;
  int rc = tail.solve_device(st);
  if (rc) return rc;
  hipLaunchKernelGGL(tops_u_kernel, dim3((unsigned)((nT + 3) / 4 + (kt + 255) / 256)), dim3(256), 0, st, nT, tk_cp, tk_ri, tk_v, tail.vin, zT, DT, uT, xext + nT,
                     kt);                                                                                        // D_T^-1 z_T - L_KT^T x_K;  x_K beside x_T
  hipLaunchKernelGGL(tops_gemv_kernel, dim3((unsigned)((nT + 3) / 4)), dim3(256), 0, st, nT, wb_off, (const int*)nullptr, wb_len, Wb, uT, xext, 0);  // x_T = W_T^T (...)
  if (tops_refine) {                                                                                             // x_T += W_T^T (u - L_TT^T x_T); zext's T part is free by now
    hipLaunchKernelGGL(tops_resid_kernel, dim3((unsigned)(((long long)nT * 32 + 255) / 256)), dim3(256), 0, st, nT, tt_cp, tt_ri, tt_cv, xext, uT, zext);
    hipLaunchKernelGGL(tops_gemv_kernel, dim3((unsigned)((nT + 3) / 4)), dim3(256), 0, st, nT, wb_off, (const int*)nullptr, wb_len, Wb, zext, xext, 1);
  }
  // w = [L_TB; L_KB]^T [x_T; x_K] -- and [x_T | x_K] into y in the caller's order on the way; the backward sweeps then write x_B straight into y
  // (round 6: the scatter kernel behind them is gone when there are leading columns to sweep)
  const bool direct = launch_l21t(n1, tptr, tri, tv_, xext, wvec, n_long, long_cols_d, st, y, kext, rid + n1);
  launch_sweeps(*this, true, ax, asmc, b, isig, xp, st, direct ? y : nullptr);                                  // x_B
  if (!direct) hipLaunchKernelGGL(tops_scatter_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, m, n1, rid, xp, xext, y);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

}  // namespace cuadmm
