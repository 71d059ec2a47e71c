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
#include <cstdio>
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
                                                          int max_levels, const int* __restrict__ tree_ids) {
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
        const double xi = lead_rhs(ax, asmc, b, isig, i) - s;
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

// w[j] = sum over the TAIL rows of column j:  L21[i][j] x2[i]   (8 lanes per leading column; independent of the sweeps)
__global__ __launch_bounds__(256) void lead_l21t_kernel(int n1, const long long* __restrict__ tp, const int* __restrict__ tr, const double* __restrict__ tv,
                                                        const double* __restrict__ x2, double* __restrict__ w) {
  const int gt = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  const int j = gt >> 3, sub = gt & 7;
  if (j >= n1) return;
  double s = 0.0;
  for (long long q = tp[j] + sub; q < tp[j + 1]; q += 8) s += tv[q] * x2[tr[q]];
  s += __shfl_xor(s, 4, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 1, 64);
  if (sub == 0) w[j] = s;
}

// backward, levels root side first:  x[j] = x[j] / D[j] - w[j] - sum_{leading i > j} L11[i][j] x[i]
__global__ __launch_bounds__(64) void lead_backward_kernel(const int* __restrict__ lvl_ptr, const int* __restrict__ lvl_off, const int* __restrict__ lvl_g,
                                                           const int* __restrict__ nodes, const long long* __restrict__ ptr, const int* __restrict__ ci,
                                                           const double* __restrict__ v, const double* __restrict__ D, const double* __restrict__ w,
                                                           double* __restrict__ x, int max_nodes, int max_levels, const int* __restrict__ tree_ids) {
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
                                                    const double* __restrict__ D, const double* __restrict__ w, double* __restrict__ x) {
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
      r_rhs[u] = i < cnt ? (BACKWARD ? x[nd] / D[nd] - w[nd] : lead_rhs(ax, asmc, b, isig, nd)) : 0.0;
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
    rhs[i] = BACKWARD ? x[nd] / D[nd] - w[nd] : lead_rhs(ax, asmc, b, isig, nd);
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
  for (int i = tid; i < cnt; i += NT) x[snode[i]] = xs[i];
}

template <bool BACKWARD, int NW>
__global__ __launch_bounds__(64 * NW) void lead_sweep_lds_kernel(const LeadTreeDesc* __restrict__ desc, const int* __restrict__ lvl_off,
                                                            const int* __restrict__ lvl_g, const int* __restrict__ nodes, const long long* __restrict__ ptr,
                                                            const int* __restrict__ ci, const double* __restrict__ v,
                                                            const double* __restrict__ ax, const double* __restrict__ asmc, const double* __restrict__ b, double isig,
                                                            const double* __restrict__ D, const double* __restrict__ w, double* __restrict__ x) {
  extern __shared__ double lead_smem[];
  lead_sweep_lds_body<BACKWARD, NW>(desc[blockIdx.x], lead_smem, (int)threadIdx.x, lvl_off, lvl_g, nodes, ptr, ci, v, ax, asmc, b, isig, D, w, x);
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
                                                             const double* __restrict__ D, const double* __restrict__ w, double* __restrict__ x) {
  extern __shared__ double lead_smem[];
  const int blk = (int)blockIdx.x;
  if (blk < n_big) {
    lead_sweep_lds_body<BACKWARD, 4>(desc_big[blk], lead_smem, (int)threadIdx.x, lvl_off, lvl_g, nodes, ptr, ci, v, ax, asmc, b, isig, D, w, x);
  } else {
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int t = 4 * (blk - n_big) + wave;
    if (t < n_small)
      lead_sweep_lds_body<BACKWARD, 1>(desc_small[t], lead_smem + (size_t)wave * small_doubles, (int)threadIdx.x & 63, lvl_off, lvl_g, nodes, ptr, ci, v, ax, asmc, b, isig,
                                       D, w, x);
  }
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
  hipLaunchKernelGGL(lead_l21t_kernel, dim3((unsigned)(((long long)n1 * 8 + 255) / 256)), dim3(256), 0, st, n1, tptr, tri, tv_, tail.vin, wvec);
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
  for (int j = 0; j < n1; ++j) {
    long long q = tp[j];
    for (long long p = Lp[j]; p < Lp[j + 1]; ++p)
      if (Li[p] >= n1) { tr[(size_t)q] = Li[p] - n1; tv[(size_t)q] = Lx[p]; ++q; }
    tp[(size_t)j + 1] = q;
  }
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
          (rc_ = to_device(tptr, tp)) || (rc_ = to_device(tri, tr)) || (rc_ = to_device(tv_, tv)) || (rc_ = hybrid_buffers(*this))) { release(); return rc_; }
      hybrid = true;
    }
    return CUADMM_OK;
  }
  lds_bytes = sizeof(double) * (size_t)max_nodes + sizeof(int) * (2 * (size_t)max_levels + 2);
  // classes by the LDS a tree needs with its stream resident (the larger of the two sweeps): small trees share a CU in numbers,
  // the few big ones get a launch of their own, anything beyond one workgroup's LDS keeps the streaming kernels
  std::vector<int> t_stream;
  std::vector<LeadTreeDesc> dsf, dsb, dbf, dbb;        // small / big trees, forward / backward sweep
  lds_small = lds_big = 0;
  {
    int acc = 0;
    std::vector<int> cnt((size_t)ntrees, 0);
    for (int j = 0; j < n1; ++j) cnt[tree_of[j]]++;
    for (int t = 0; t < ntrees; ++t) {
      const long long nzf = fp[(size_t)acc + cnt[t]] - fp[acc], nzb = bp[(size_t)acc + cnt[t]] - bp[acc];
      const int nlf = lpf[(size_t)t + 1] - lpf[t] - 1, nlb = lpb[(size_t)t + 1] - lpb[t] - 1;
      const long long nz = std::max(nzf, nzb);
      const int nlev_t = std::max(nlf, nlb);
      const size_t need = sizeof(double) * (2 * (size_t)cnt[t] + (size_t)nz) + sizeof(int) * (2 * (size_t)cnt[t] + 1 + (size_t)nz + 2 * (size_t)nlev_t + 2) + 16;
      const LeadTreeDesc df{lpf[t], nlf, acc, cnt[t], (int)nzf, 0, fp[acc]}, db{lpb[t], nlb, acc, cnt[t], (int)nzb, 0, bp[acc]};
      if (stream_only || need > kMaxLdsBytes - 1024) t_stream.push_back(t);
      else if (need <= 16 * 1024) { dsf.push_back(df); dsb.push_back(db); lds_small = std::max(lds_small, need); }
      else { dbf.push_back(df); dbb.push_back(db); lds_big = std::max(lds_big, need); }
      acc += cnt[t];
    }
  }
  n_small = (int)dsf.size(); n_big = (int)dbf.size(); n_stream = (int)t_stream.size();
  if (debug) fprintf(stderr, "[lead debug] resident trees: %d small (%zu B), %d big (%zu B), %d streaming\n", n_small, lds_small, n_big, lds_big, n_stream);
  int rc;
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
      (rc = to_device(tptr, tp)) || (rc = to_device(tri, tr)) || (rc = to_device(tv_, tv)) ||
      (rc = to_device(D1, std::vector<double>(D, D + n1))) || (rc = to_device(nodes_f, nf)) || (rc = to_device(nodes_b, nb)) ||
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

int LeadSolve::solve(const double* ax, const double* asmc, const double* b, double isig, double* y, TailSolve& tail, hipStream_t st) const {
  if (!ready) { set_error("lead_solve: not built"); return CUADMM_ERR_INVALID; }
  // big trees (four wavefronts each) and small ones (one wavefront each, four per workgroup) in ONE launch when both exist
  const bool merged = n_big > 0 && n_small > 0;
  const int small_doubles = (int)((lds_small + 7) / 8);
  const size_t lds_merged = std::max(lds_big, 4 * sizeof(double) * (size_t)small_doubles);
  const unsigned grid_merged = (unsigned)(n_big + (n_small + 3) / 4);
  if (merged)
    hipLaunchKernelGGL(lead_sweep_merged_kernel<false>, dim3(grid_merged), dim3(256), lds_merged, st, static_cast<const LeadTreeDesc*>(desc_big_f), n_big,
                       static_cast<const LeadTreeDesc*>(desc_small_f), n_small, small_doubles, lvl_off_f, lvl_g_f, nodes_f, fptr, fci, fv_, ax, asmc, b, isig,
                       (const double*)nullptr, (const double*)nullptr, y);
  else if (n_small > 0)
    hipLaunchKernelGGL((lead_sweep_lds_kernel<false, 1>), dim3(n_small), dim3(64), lds_small, st, static_cast<const LeadTreeDesc*>(desc_small_f), lvl_off_f,
                       lvl_g_f, nodes_f, fptr, fci, fv_, ax, asmc, b, isig, (const double*)nullptr, (const double*)nullptr, y);
  else if (n_big > 0)
    hipLaunchKernelGGL((lead_sweep_lds_kernel<false, 4>), dim3(n_big), dim3(256), lds_big, st, static_cast<const LeadTreeDesc*>(desc_big_f), lvl_off_f, lvl_g_f,
                       nodes_f, fptr, fci, fv_, ax, asmc, b, isig, (const double*)nullptr, (const double*)nullptr, y);
  if (n_stream > 0) hipLaunchKernelGGL(lead_forward_kernel, dim3(n_stream), dim3(64), lds_bytes, st, lvl_ptr_f, lvl_off_f, lvl_g_f, nodes_f, fptr, fci, fv_, ax, asmc, b,
                                       isig, y, max_nodes, max_levels, trees_stream);
  hipLaunchKernelGGL(lead_tail_rhs_kernel, dim3((k * 8 + 255) / 256), dim3(256), 0, st, k, n1, rp21, ci21, v21, ax, asmc, b, isig, y, tail.vin);
  CUADMM_HIP_TRY(hipGetLastError());
  int rc = tail.solve_device(st);                    // vin <- L22^-T D2^-1 L22^-1 vin (padding beyond k stays zero)
  if (rc) return rc;
  if (n1 > 0) hipLaunchKernelGGL(lead_l21t_kernel, dim3((n1 * 8 + 255) / 256), dim3(256), 0, st, n1, tptr, tri, tv_, tail.vin, wvec);
  if (merged)
    hipLaunchKernelGGL(lead_sweep_merged_kernel<true>, dim3(grid_merged), dim3(256), lds_merged, st, static_cast<const LeadTreeDesc*>(desc_big_b), n_big,
                       static_cast<const LeadTreeDesc*>(desc_small_b), n_small, small_doubles, lvl_off_b, lvl_g_b, nodes_b, bptr, bci, bv_,
                       (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, 0.0, D1, wvec, y);
  else if (n_small > 0)
    hipLaunchKernelGGL((lead_sweep_lds_kernel<true, 1>), dim3(n_small), dim3(64), lds_small, st, static_cast<const LeadTreeDesc*>(desc_small_b), lvl_off_b,
                       lvl_g_b, nodes_b, bptr, bci, bv_, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, 0.0, D1, wvec, y);
  else if (n_big > 0)
    hipLaunchKernelGGL((lead_sweep_lds_kernel<true, 4>), dim3(n_big), dim3(256), lds_big, st, static_cast<const LeadTreeDesc*>(desc_big_b), lvl_off_b, lvl_g_b,
                       nodes_b, bptr, bci, bv_, (const double*)nullptr, (const double*)nullptr, (const double*)nullptr, 0.0, D1, wvec, y);
  if (n_stream > 0) hipLaunchKernelGGL(lead_backward_kernel, dim3(n_stream), dim3(64), lds_bytes, st, lvl_ptr_b, lvl_off_b, lvl_g_b, nodes_b, bptr, bci, bv_, D1, wvec, y,
                                       max_nodes, max_levels, trees_stream);
  CUADMM_HIP_TRY(hipGetLastError());
  // the solved tail into y: a kernel of its own, not hipMemcpyAsync -- the runtime's device-to-device copy is a blit behind ~15 us of
  // command-processor work (kernel trace of pendulum N = 80: 14.7 us idle in front of every one, once per solve)
  hipLaunchKernelGGL(lead_copy_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, st, tail.vin, y + n1, k);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

}  // namespace cuadmm
