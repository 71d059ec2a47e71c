// PSD-cone projection kernels (gfx950) and their host-side planner/launcher.
//
// Projection (MODE 0), one launch per size class:
//   n <= 16              : psd_small_reg_kernel<4|8|16>: register-resident eigensolver, 16..4 blocks per wavefront
//   9 <= n <= 64         : psd_sign_wave_kernel / psd_sign_closed_kernel: matrix-sign iteration on the fp64 matrix cores, one wavefront per block
//   33 <= n <= 64        : psd_sign_lds_kernel<48|64>: same iteration resident in LDS, one workgroup per block
//   larger               : psd_large.hip (batched GEMM launches)
// The sign kernels stop per block (sign_sched.h).  Explicit eigendecomposition (MODE 1, cuadmm_op_batch_eig):
//   n <= 64              : psd_small_reg_kernel<NMAX>
//   65 <= n <= ~136      : psd_wg_kernel<NT>: one workgroup per block, matrix in LDS (op entry: n <= 128)
//   larger               : eig_large.hip: one matrix at a time on the whole chip (also the rank-limited projection of such blocks)
// MODE 0: svec in -> projected svec out (the fused replacement of solver.cu:534-647)
// MODE 1: dense column-major symmetric in -> eigenvectors (column-major) + ascending eigenvalues
//         (the contract of the reference's cuSOLVER wrappers, cusolver.h:76-95,154-171)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "device_util.h"
#include "eig_large.h"
#include "psd_device.h"
#include "psd_plan.h"
#include "psd_small_reg.h"
#include "psd_sign_lds.h"
#include "psd_sign_wave.h"
#include "psd_sign_closed.h"

namespace cuadmm {

struct PsdArgs {
  const double* in;
  double* out;
  double* Wout;            // MODE 1
  int* info;               // MODE 1: per matrix flag; MODE 0: single counter (may be null)
  const int* ids;          // class member -> block id (null: identity)
  const long long* boff;   // svec offset per block (MODE 0)
  const int* bn;           // size per block (MODE 0)
  int count;               // members in this launch
  int n_uniform;           // MODE 1
  long long* dbg;          // developer aid: per-workgroup phase timestamps (CUADMM_PSD_DEBUG)
  int* steps;              // developer aid: Newton-Schulz steps taken per block (sign kernels; may be null)
  int* hint;               // per block, in/out: lift steps the previous projection needed (schedule warm start; may be null)
  int eig_rank;            // > 0: rank-limited projection, only the eig_rank largest eigenvalues survive (eigensolver kernels)
  const PsdDesc* desc;     // class member -> (svec offset, size, block id) in ONE 16-byte load (the sign kernels' first dependent access)
};

// register-resident variant (psd_small_reg.h): the production path for n <= 32
template <int NMAX, int MODE>
__global__ __launch_bounds__(64) void psd_small_reg_kernel(PsdArgs a) {
  __shared__ double smem[(64 / NMAX) * RegLayout<NMAX>::kPer];
  psd_small_reg_body<NMAX, MODE>(a, smem);
  if (MODE == 0) {
    wave_fence();
    const int slot0 = (int)blockIdx.x * (64 / NMAX);
    long long* dbg = a.dbg ? a.dbg + (long long)blockIdx.x * 8 : nullptr;
    if constexpr (NMAX >= 16) {
      const long long t0 = dbg ? (long long)__builtin_readcyclecounter() : 0;
      psd_small_reg_rebuild_mfma<NMAX, RegLayout<NMAX>>(a, smem, slot0);
      if (dbg && lane_id() == 0) dbg[6] = (long long)__builtin_readcyclecounter() - t0;
    }
    psd_small_reg_store<NMAX, RegLayout<NMAX>>(a, smem, slot0, dbg);
  }
}

template <int NT, int MODE>
__global__ __launch_bounds__(NT) void psd_wg_kernel(PsdArgs a) {
  extern __shared__ __attribute__((aligned(16))) double dsm[];
  using Gp = WgGroup<NT>;
  constexpr int NW = NT / 64;
  const int slot = (int)blockIdx.x;
  const int bi = a.ids ? a.ids[slot] : slot;
  const int n = (MODE == 0) ? a.bn[bi] : a.n_uniform;
  const int ld = n | 1;
  double* M = dsm;
  double* vecs = dsm + (size_t)n * ld;
  double* dsh = vecs;
  double* esh = dsh + n;
  double* tau = esh + n;
  double* vv = tau + n;
  double* ww = vv + n;
  double* scratch = ww + n;                 // NW doubles (padded to 8)
  double* dq = (NW > 1) ? scratch + 8 + (size_t)(threadIdx.x >> 6) * 2 * n : dsh;
  double* eq = (NW > 1) ? dq + n : esh;

  const int tid = (int)threadIdx.x;
  if (MODE == 0) {
    const double* src = a.in + a.boff[bi];
    const int len = n * (n + 1) / 2;
    for (int e = tid; e < len; e += NT) {
      int i, j;
      tri_decode(e, i, j);
      double v = src[e];
      if (i != j) v *= kSqrt2Inv;
      M[(size_t)j * ld + i] = v;
      M[(size_t)i * ld + j] = v;
    }
  } else {
    const double* src = a.in + (long long)bi * n * n;
    for (int idx = tid; idx < n * n; idx += NT) {
      const int c = idx / n, r = idx - c * n;
      if (r >= c) {
        const double v = src[idx];
        M[(size_t)r * ld + c] = v;
        M[(size_t)c * ld + r] = v;
      }
    }
  }
  __syncthreads();
  const int fail = sym_eig_inplace<Gp>(M, ld, n, dsh, esh, tau, vv, ww, dq, eq, scratch);
  if (MODE == 0) {
    reconstruct_to_svec<Gp>(M, ld, n, dq, vv, a.out + a.boff[bi], a.eig_rank);
    if (fail && tid == 0 && a.info) atomicAdd(a.info, 1);
  } else {
    write_sorted_eig<Gp>(M, ld, n, dq, a.out + (long long)bi * n * n, a.Wout + (long long)bi * n);
    if (tid == 0 && a.info) a.info[bi] = fail;
  }
}

// unconstrained ('u') blocks: the projection onto R^n is the identity -- one workgroup per block copies its svec range
__global__ __launch_bounds__(256) void copy_ranges_kernel(const double* __restrict__ in, double* __restrict__ out,
                                                          const long long* __restrict__ off, const long long* __restrict__ len) {
  const double* s = in + off[blockIdx.x];
  double* d = out + off[blockIdx.x];
  for (long long i = threadIdx.x; i < len[blockIdx.x]; i += 256) d[i] = s[i];
}

// ---------------------------------------------------------------------------------------
// planner
// ---------------------------------------------------------------------------------------
// 32 < n <= 64 (projection only): matrix-sign iteration resident in LDS, one workgroup per block (psd_sign_lds.h)
template <int NP, bool TRIPLE>
__global__ __launch_bounds__(SignLdsCfg<NP>::THREADS) void psd_sign_lds_kernel(PsdArgs a, int first) {
  extern __shared__ double sign_smem[];
  const int m = first + (int)blockIdx.x;
  const int id = a.ids ? a.ids[m] : m;
  psd_sign_lds_body<NP, TRIPLE>(a.in + a.boff[id], a.out + a.boff[id], a.bn[id], a.info, sign_smem, a.steps ? a.steps + id : nullptr,
                        a.hint ? a.hint + id : nullptr);
}

// one WAVEFRONT per block (psd_sign_wave.h): NT = 1 (n <= 16, eight wavefronts per SIMD), NT = 2 (n <= 32, four -- or three: option
// psd_w32_occ), NT = 3 (n <= 48, two), NT = 4 (n <= 64, one)
template <int NT, int OCC, bool FUSED>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void psd_sign_wave_kernel(PsdArgs a, SignFuse fz, int first, int count) {
  extern __shared__ double swt_smem[];
  const int m = (int)blockIdx.x;
  if (m >= count) return;
  const PsdDesc d = a.desc[first + m];
  const int id = d.id;
  const long long off = d.off;
  psd_sign_wave_body<NT, FUSED, !(NT == 1 && OCC == 8)>(a.in + off, a.out + off, d.n, a.info, swt_smem, a.steps ? a.steps + id : nullptr,
                                a.hint ? a.hint + id : nullptr, a.dbg ? a.dbg + 10 * (long long)m : nullptr, fz, off, d.slot, id);
}

// closed blocks (psd_sign_closed.h): the whole iteration of the block, one launch per iteration
template <int NT, int OCC, bool FULL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void psd_sign_closed_kernel(ClosedArgs a) {
  extern __shared__ double swt_smem[];
  const int m = (int)blockIdx.x;
  if (m >= a.count) return;
  const PsdDesc d = a.desc[a.first + m];
  // the epilogue re-reads the arguments from the kernarg segment (psd_sign_closed.h: swc_args) instead of keeping them in scalar
  // registers across the iteration
  SwcKArg ka = (SwcKArg)__builtin_amdgcn_kernarg_segment_ptr();
  psd_sign_closed_body<NT, false, FULL>(a, d.n, swt_smem, a.steps ? a.steps + d.id : nullptr, a.hint ? a.hint + d.id : nullptr, nullptr, d.off, d.slot, 0, d.pad[0], 0, ka);
}

// SEVERAL ADMM ITERATIONS PER LAUNCH (ClosedArgs::iters; closed blocks): one PERSISTENT WORKGROUP PER CU (WAVES = the CU's
// wavefront slots at this kernel's register budget; n <= 16: two workgroups of 16) owns a fixed contiguous range of the class
// members and runs the tasks (iteration it, member j), in the order it * nb + j, on whichever of its wavefronts is free: an LDS
// counter hands them out, an LDS array `done[j]` = iterations completed on member j orders a task behind its predecessor (the
// same member, one iteration earlier: nb tasks back in the queue with at most WAVES in flight, so the wait is almost never
// taken).  What task (it, j) reads -- X, S, y and the member's rows of [A X | A (S - C)] -- was stored by a wavefront of the
// SAME CU (same L1): a workgroup-scope release (s_waitcnt vmcnt(0)) before the flag and an acquire after it order the two.
// Why: with one wavefront per block and one launch per iteration, 10 000 blocks on 4 096 wavefront slots are 2.44 rounds -- the
// last one at 44 % occupancy -- plus a launch ramp, a tail and a host round trip every iteration; handing out (member,
// iteration) pairs dynamically inside a CU keeps every SIMD at its four wavefronts until the last task of the batch.
// ints of the task counter + the members' flags, rounded so that the descriptor cache behind them is 16-byte aligned
__host__ __device__ constexpr int closed_cu_ctl_ints(int nb_max) { return (nb_max + 2 + 3) & ~3; }
static_assert(sizeof(PsdDesc) == 32, "the LDS descriptor cache copies a descriptor as two 16-byte words");

template <int NT, int WAVES, int OCC, bool FULL, bool DBG>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void psd_sign_closed_cu_kernel(ClosedArgs a) {
  extern __shared__ double swt_smem[];
  constexpr int TILE = SignWaveT<NT>::NP * SignWaveT<NT>::LD;
  int* ctl = reinterpret_cast<int*>(swt_smem + (size_t)WAVES * TILE);      // [0]: next task; [1 + j]: iterations completed on member j
  // members g, g + G, g + 2 G, ... : the class is sorted longest block first, so a STRIDED share gives every workgroup the same mix
  // of long and short blocks (contiguous shares left the first workgroups with ~20 % more Newton-Schulz steps than the last)
  const int g = (int)blockIdx.x, G = (int)gridDim.x;
  const int nb = (a.count - g + G - 1) / G;
  for (int i = (int)threadIdx.x; i < nb + 1; i += 64 * WAVES) ctl[i] = 0;
  // The descriptors of the workgroup's members, copied to LDS once (a task's descriptor does not depend on its iteration): the
  // first of a task's dependent memory round trips -- a global load of 3-6 k ticks on the loaded chip -- becomes an LDS read.
  sl_v4i32* dcache = reinterpret_cast<sl_v4i32*>(ctl + closed_cu_ctl_ints((a.count + G - 1) / G));
  if (a.dcache)
    for (int i = (int)threadIdx.x; i < nb; i += 64 * WAVES) {
      const sl_v4i32* src = reinterpret_cast<const sl_v4i32*>(a.desc + a.first + g + i * G);
      dcache[2 * i] = src[0];
      dcache[2 * i + 1] = src[1];
    }
  __syncthreads();
  const int ntask = nb * a.iters;
  const long long clk0 = (DBG && a.dbg) ? (long long)__builtin_readcyclecounter() : 0, rt0 = (DBG && a.dbg) ? (long long)__builtin_amdgcn_s_memrealtime() : 0;
  // The kernel arguments are re-read from the kernarg segment for every task (scalar loads, cached): held in SGPRs across the
  // task loop they would be spilled at this kernel's register budget.
  using KArgC = __attribute__((address_space(4))) const char;
  KArgC* ka = (KArgC*)__builtin_amdgcn_kernarg_segment_ptr();
#pragma unroll 1
  for (;;) {
    const long long tk0 = (DBG && a.dbg) ? (long long)__builtin_readcyclecounter() : 0;
    int t = 0;
    if (lane_id() == 0) t = atomicAdd(&ctl[0], 1);
    t = __builtin_amdgcn_readfirstlane(t);
    if (t >= ntask) break;
    const int it = t / nb, j = t - it * nb;
    if (it > 0) {   // plain LDS atomics: a volatile access would become a FLAT one (system scope, a vmcnt(0) round trip each)
      while (__hip_atomic_load(ctl + 1 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < it) __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const long long tk1 = (DBG && a.dbg) ? (long long)__builtin_readcyclecounter() : 0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(ka));                       // opaque: nothing loaded through it is kept across tasks
    const ClosedArgs al = *reinterpret_cast<__attribute__((address_space(4))) const ClosedArgs*>(ka);
#else
    const ClosedArgs al = a;
    (void)ka;
#endif
    PsdDesc d;
    if (al.dcache) {
      const sl_v4i32 q0 = dcache[2 * j], q1 = dcache[2 * j + 1];
      d.off = (long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane(q0[1]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane(q0[0]));
      d.n = __builtin_amdgcn_readfirstlane(q0[2]); d.id = __builtin_amdgcn_readfirstlane(q0[3]); d.slot = __builtin_amdgcn_readfirstlane(q1[0]);
      d.pad[0] = __builtin_amdgcn_readfirstlane(q1[1]);
    } else {
      d = al.desc[al.first + g + j * G];
    }
    int toff = ((int)threadIdx.x >> 6) * TILE;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(toff));                     // not hoisted out of the task loop (it would stay live across the body)
#endif
    double* tile = swt_smem + toff;
    const long long tk2 = (DBG && al.dbg) ? (long long)__builtin_readcyclecounter() : 0;
    psd_sign_closed_body<NT, true, FULL, DBG>(al, d.n, tile, al.steps ? al.steps + d.id : nullptr, al.hint ? al.hint + d.id : nullptr,
                             (DBG && al.dbg) ? al.dbg + 16 * (long long)(g + j * G) : nullptr, d.off, d.slot, (long long)it * al.pstride, d.pad[0], it, ka);
    const long long tk3 = (DBG && al.dbg) ? (long long)__builtin_readcyclecounter() : 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane_id() == 0) __hip_atomic_store(ctl + 1 + j, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (DBG && al.dbg && lane_id() == 0) {   // developer aid (psd_debug = 2): what a task spends outside its body
      long long* q = al.dbg + 16 * (long long)(g + j * G);
      q[10] = tk1 - tk0; q[11] = tk2 - tk1; q[12] = tk3 - tk2; q[13] = (long long)__builtin_readcyclecounter() - tk3;
    }
  }
  if (DBG && a.dbg && threadIdx.x == 0) {   // the shader clock over the launch: s_memtime ticks per 100 MHz real-time tick
    long long* q = a.dbg + 16 * (long long)g;
    q[14] = (long long)__builtin_readcyclecounter() - clk0; q[15] = (long long)__builtin_amdgcn_s_memrealtime() - rt0;
  }
}

// one persistent launch; the LDS cap is raised once per kernel and device
template <void (*KERN)(ClosedArgs)>
static int launch_closed_cu(dim3 grid, dim3 block, size_t lds, hipStream_t st, const ClosedArgs& ca) {
  static LdsCapOnce once;
  CUADMM_HIP_TRY(once(reinterpret_cast<const void*>(KERN)));
  hipLaunchKernelGGL(KERN, grid, block, lds, st, ca);
  return CUADMM_OK;
}

// fz != nullptr: the fused variant (SignFuse, psd_sign_wave.h); the partial-sum slot of a member comes with its descriptor
template <int NT, int OCC>
static int launch_sign_wave(const PsdArgs& a, int first, int count, hipStream_t st, const SignFuse* fz = nullptr) {
  if (count <= 0) return CUADMM_OK;
  if (fz && fz->rec) {                                   // closed blocks: psd_sign_closed.h
    ClosedArgs ca{};
    ca.desc = a.desc; ca.steps = a.steps; ca.hint = a.hint; ca.fail = a.info; ca.dbg = a.dbg;
    ca.X = fz->X; ca.S = fz->S; ca.Rd1 = fz->Rd1; ca.C = fz->C; ca.rec = fz->rec; ca.cl_out = fz->cl_out; ca.y_out = fz->y_out;
    ca.outS = fz->outS; ca.outX = fz->outX; ca.partials = fz->partials; ca.partials2 = fz->partials2;
    ca.sig = fz->sig; ca.inv_sig = fz->inv_sig; ca.tau_sig = fz->tau_sig; ca.isig = fz->isig; ca.bscale = fz->bscale;
    ca.iter0 = fz->iter0;
    ca.pstride = fz->pstride; ca.mode = fz->mode; ca.iters = fz->iters > 1 ? fz->iters : 1; ca.first = first; ca.count = count;
    if (fz->iters > 1) {
      constexpr int WAVES = 4 * OCC > 16 ? 16 : 4 * OCC, WG_PER_CU = 4 * OCC / WAVES;
      static int n_cu_of[64] = {0};                    // per DEVICE (two solvers of a process may sit on different GPUs)
      int dev = 0;
      CUADMM_HIP_TRY(hipGetDevice(&dev));
      int& n_cu = n_cu_of[dev & 63];
      if (n_cu == 0) {
        int v = 0;
        CUADMM_HIP_TRY(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
        n_cu = v > 0 ? v : 256;
      }
      const int grid = std::min(count, n_cu * WG_PER_CU);
      const size_t nb_max = ((size_t)count + grid - 1) / grid;
      size_t lds = WAVES * SignWaveT<NT>::LDS_BYTES + sizeof(int) * (size_t)closed_cu_ctl_ints((int)nb_max);
      ca.dcache = lds + sizeof(PsdDesc) * nb_max <= kMaxLdsBytes;     // the members' descriptors in LDS when they fit
      if (ca.dcache) lds += sizeof(PsdDesc) * nb_max;
      if (lds > kMaxLdsBytes) { set_error("psd: %d blocks per workgroup do not fit the batched launch", (int)nb_max); return CUADMM_ERR_INVALID; }
      constexpr bool HAS_DBG = NT == 2 && OCC == 4;     // the tick stamps exist for the C2 geometry only
      const dim3 gr(grid), bl(64 * WAVES);
      int rc;
      if (HAS_DBG && ca.dbg) rc = fz->full ? launch_closed_cu<psd_sign_closed_cu_kernel<NT, WAVES, OCC, true, HAS_DBG>>(gr, bl, lds, st, ca)
                                           : launch_closed_cu<psd_sign_closed_cu_kernel<NT, WAVES, OCC, false, HAS_DBG>>(gr, bl, lds, st, ca);
      else rc = fz->full ? launch_closed_cu<psd_sign_closed_cu_kernel<NT, WAVES, OCC, true, false>>(gr, bl, lds, st, ca)
                         : launch_closed_cu<psd_sign_closed_cu_kernel<NT, WAVES, OCC, false, false>>(gr, bl, lds, st, ca);
      if (rc) return rc;
    } else if (fz->full) {
      hipLaunchKernelGGL((psd_sign_closed_kernel<NT, OCC, true>), dim3(count), dim3(64), SignWaveT<NT>::LDS_BYTES, st, ca);
    } else {
      hipLaunchKernelGGL((psd_sign_closed_kernel<NT, OCC, false>), dim3(count), dim3(64), SignWaveT<NT>::LDS_BYTES, st, ca);
    }
  }
  else {
    // n <= 16 at eight wavefronts per SIMD is a 64-register kernel: with the mega-lift's state it spills (18 VGPRs to scratch).  Eight
    // per SIMD only matter when the class fills the chip; a moment relaxation's few hundred blocks run the 128-register instantiation.
    constexpr int OCC_SMALL = (NT == 1 && OCC == 8) ? 4 : OCC;
    const bool small_class = OCC_SMALL != OCC && count <= 4096;
    if (fz) {
      if (small_class) hipLaunchKernelGGL((psd_sign_wave_kernel<NT, OCC_SMALL, true>), dim3(count), dim3(64), SignWaveT<NT>::LDS_BYTES, st, a, *fz, first, count);
      else hipLaunchKernelGGL((psd_sign_wave_kernel<NT, OCC, true>), dim3(count), dim3(64), SignWaveT<NT>::LDS_BYTES, st, a, *fz, first, count);
    } else {
      if (small_class) hipLaunchKernelGGL((psd_sign_wave_kernel<NT, OCC_SMALL, false>), dim3(count), dim3(64), SignWaveT<NT>::LDS_BYTES, st, a, SignFuse{}, first, count);
      else hipLaunchKernelGGL((psd_sign_wave_kernel<NT, OCC, false>), dim3(count), dim3(64), SignWaveT<NT>::LDS_BYTES, st, a, SignFuse{}, first, count);
    }
  }
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

// 17 <= n <= 32: three or four wavefronts per SIMD (PsdOptions::w32_occ; the launches of several iterations: cu_occ)
static int launch_sign_wave32(const PsdArgs& a, int first, int count, hipStream_t st, const PsdOptions& opt, const SignFuse* fz = nullptr) {
  if (count <= 0) return CUADMM_OK;
  const int occ = (fz && fz->iters > 1) ? opt.cu_occ : opt.w32_occ;
  if (occ == 3) return launch_sign_wave<2, 3>(a, first, count, st, fz);
  return launch_sign_wave<2, 4>(a, first, count, st, fz);
}

template <int NP, bool TRIPLE>
static int launch_sign_lds_t(const PsdArgs& a, int first, int count, hipStream_t st) {
  const size_t lds = sizeof(double) * (TRIPLE ? 3 : 2) * NP * SignLdsCfg<NP>::LD;
  auto kern = psd_sign_lds_kernel<NP, TRIPLE>;
  static LdsCapOnce once;
  if (lds > 48 * 1024) CUADMM_HIP_TRY(once(reinterpret_cast<const void*>(kern)));
  hipLaunchKernelGGL(kern, dim3(count), dim3(SignLdsCfg<NP>::THREADS), lds, st, a, first);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}
// triple = the variant with a third matrix in LDS (one barrier and the statistics less on most steps): when the class leaves at
// most one workgroup per CU anyway (a moment relaxation's handful of blocks), so the extra LDS costs no occupancy
template <int NP>
static int launch_sign_lds(const PsdArgs& a, int first, int count, hipStream_t st, bool triple) {
  if (count <= 0) return CUADMM_OK;
  return triple ? launch_sign_lds_t<NP, true>(a, first, count, st) : launch_sign_lds_t<NP, false>(a, first, count, st);
}

static size_t wg_lds_bytes(int n, int nt) {
  const int ld = n | 1;
  const int nw = nt / 64;   // a single wavefront runs QL directly on the shared d/e
  size_t doubles = (size_t)n * ld + 5 * (size_t)n + 8 + (nw > 1 ? 2 * (size_t)n * nw : 0);
  return doubles * sizeof(double);
}

static int wg_threads_lds(int n) { return n <= 64 ? 64 : (n <= 128 ? 128 : 256); }

int psd_class_of(int n) {
  if (n <= 4) return 0;
  if (n <= 8) return 1;
  if (n <= 16) return 2;
  if (n <= 32) return 3;
  if (n <= 64) return 4;
  if (wg_lds_bytes(n, wg_threads_lds(n)) <= kMaxLdsBytes) return 5;
  return 6;
}

int PsdPlan::build(const int* blk, int mat_num) {
  release();
  nblk = mat_num;
  h_blk.assign(blk, blk + mat_num);
  sign16 = opt.n16_sign != 0;
  sign.opt = opt;
  std::vector<long long> off((size_t)mat_num + 1, 0);
  std::vector<long long> free_off, free_len;   // unconstrained blocks (negative size): identity "projection"
  for (int k = 0; k < mat_num; ++k) {
    if (blk[k] == 0) { set_error("block %d has size 0", k); return CUADMM_ERR_INVALID; }
    if (blk[k] > kMaxBlockSize) { set_error("block %d has size %d > %d (largest supported this build)", k, blk[k], kMaxBlockSize); return CUADMM_ERR_INVALID; }
    off[k + 1] = off[k] + blk_svec_len(blk[k]);
    if (blk[k] < 0) { free_off.push_back(off[k]); free_len.push_back(-(long long)blk[k]); }
  }
  vec_len = off[mat_num];
  sign_min = std::max(65, opt.sign_min);
  // a rank mask needs eigenvalues: with eig_rank set every block goes through the eigensolver kernels
  if (eig_rank > 0) sign_min = 0x7fffffff;
  std::vector<int> sign_members;
  for (int k = 0; k < mat_num; ++k)
    if (blk[k] >= sign_min) sign_members.push_back(k);
  std::vector<int> ids;
  cls4_big = 0;
  for (int c = 0; c < kNumPsdClasses; ++c) {
    cls_begin[c] = (int)ids.size();
    std::vector<int> members;
    for (int k = 0; k < mat_num; ++k)
      if (blk[k] > 0 && class_of(blk[k]) == c && blk[k] < sign_min) members.push_back(k);
    std::stable_sort(members.begin(), members.end(), [&](int x, int y) { return blk[x] > blk[y]; });
    for (int k : members) {
      ids.push_back(k);
      if (c >= 4) cls_maxn[c] = std::max(cls_maxn[c], blk[k]);
      if (c == 4 && blk[k] > 48) ++cls4_big;
    }
    cls_count[c] = (int)ids.size() - cls_begin[c];
  }
  wave4 = opt.mid == 0 && cls_count[4] >= opt.wave4_min;
  // ranges whose members all fill their tile exactly (n = 16 NT): the closed-block kernels then know every slot of the svec walks
  // at compile time (psd_sign_closed.h, FULL)
  {
    auto all_eq = [&](int begin, int count, int n) { for (int q = begin; q < begin + count; ++q) if (blk[ids[(size_t)q]] != n) return false; return count > 0; };
    range_full[0] = all_eq(cls_begin[2], cls_count[2], 16);
    range_full[1] = all_eq(cls_begin[3], cls_count[3], 32);
    range_full[2] = all_eq(cls_begin[4] + cls4_big, cls_count[4] - cls4_big, 48);
    range_full[3] = all_eq(cls_begin[4], cls4_big, 64);
  }
  n_free = (int)free_off.size();
  if (n_free > 0) {
    CUADMM_HIP_TRY(hipMalloc(&d_free_off, sizeof(long long) * free_off.size()));
    CUADMM_HIP_TRY(hipMalloc(&d_free_len, sizeof(long long) * free_len.size()));
    { int rc_ = staged_h2d(d_free_off, free_off.data(), sizeof(long long) * free_off.size()); if (rc_) return rc_; }
    { int rc_ = staged_h2d(d_free_len, free_len.data(), sizeof(long long) * free_len.size()); if (rc_) return rc_; }
  }
  CUADMM_HIP_TRY(hipMalloc(&d_off, sizeof(long long) * ((size_t)mat_num + 1)));
  CUADMM_HIP_TRY(hipMalloc(&d_n, sizeof(int) * (size_t)std::max(mat_num, 1)));
  CUADMM_HIP_TRY(hipMalloc(&d_ids, sizeof(int) * (size_t)std::max(mat_num, 1)));
  CUADMM_HIP_TRY(hipMalloc(&d_fail, sizeof(int)));
  { int rc_ = staged_h2d(d_off, off.data(), sizeof(long long) * ((size_t)mat_num + 1)); if (rc_) return rc_; }
  { int rc_ = staged_h2d(d_n, blk, sizeof(int) * (size_t)mat_num); if (rc_) return rc_; }
  if (!ids.empty()) { int rc_ = staged_h2d(d_ids, ids.data(), sizeof(int) * ids.size()); if (rc_) return rc_; }
  h_ids = ids;
  {
    std::vector<PsdDesc> desc(std::max<size_t>(ids.size(), 1));
    std::vector<int> slot_of;
    fused_slots(slot_of);
    for (size_t q = 0; q < ids.size(); ++q) desc[q] = PsdDesc{off[ids[q]], blk[ids[q]], ids[q], slot_of[ids[q]], {0, 0, 0}};
    CUADMM_HIP_TRY(hipMalloc(&d_desc, sizeof(PsdDesc) * desc.size()));
    CUADMM_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_desc_pin), sizeof(PsdDesc) * desc.size(), hipHostMallocDefault));   // reorder_by_steps_async
    { int rc_ = staged_h2d(d_desc, desc.data(), sizeof(PsdDesc) * desc.size()); if (rc_) return rc_; }
    h_desc = desc;
  }
  {
    int rc = sign.build(blk, sign_members);
    if (rc) return rc;
  }
  CUADMM_HIP_TRY(hipMemset(d_fail, 0, sizeof(int)));
  h_off = off;
  dominant_geometry = compute_dominant_geometry();
  // nominal flops 10.67 n^3 per block, GEMM-shaped part 2 n^3
  sum_n3 = 0;
  for (int k = 0; k < mat_num; ++k) if (blk[k] > 0) sum_n3 += (double)blk[k] * blk[k] * blk[k];
  return CUADMM_OK;
}

void PsdPlan::release() {
  for (void* p : {(void*)d_off, (void*)d_n, (void*)d_ids, (void*)d_fail, (void*)d_free_off, (void*)d_free_len, (void*)d_rest, (void*)d_desc})
    if (p) { hipError_t e = hipFree(p); (void)e; }
  d_rest = nullptr; n_rest = 0; d_desc = nullptr;
  if (h_desc_pin) { hipError_t e = hipHostFree(h_desc_pin); (void)e; h_desc_pin = nullptr; }
  d_off = nullptr; d_n = nullptr; d_ids = nullptr; d_fail = nullptr;
  big_ws.release();
  d_free_off = d_free_len = nullptr; n_free = 0;
  sign.release();
  if (ev_fork) {
    hipError_t e = hipEventDestroy(ev_fork); (void)e;
    for (int i = 0; i < kNumPsdClasses; ++i) { e = hipEventDestroy(ev_done[i]); (void)e; e = hipStreamDestroy(aux[i]); (void)e; }
    ev_fork = nullptr;
  }
  for (int c = 0; c < kNumPsdClasses; ++c) { cls_begin[c] = cls_count[c] = 0; cls_maxn[c] = 0; }
}

template <int NT, int MODE>
static int launch_wg(const PsdArgs& a, int maxn, hipStream_t st) {
  const size_t lds = wg_lds_bytes(maxn, NT);
  if (lds > kMaxLdsBytes) { set_error("psd: block of size %d needs %zu bytes of LDS", maxn, lds); return CUADMM_ERR_INVALID; }
  auto kern = psd_wg_kernel<NT, MODE>;
  static LdsCapOnce once;            // the hardware maximum, once per device (plans with smaller blocks must not lower it)
  if (lds > 48 * 1024) CUADMM_HIP_TRY(once(reinterpret_cast<const void*>(kern)));
  hipLaunchKernelGGL(kern, dim3(a.count), dim3(NT), lds, st, a);
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

// eigensolver kernels per size class (MODE 0: projection of the n <= 16 classes; MODE 1: explicit eigendecomposition)
template <int MODE>
static int launch_class(int c, PsdArgs a, int maxn, hipStream_t st) {
  if (a.count <= 0) return CUADMM_OK;
  switch (c) {
    case 0: hipLaunchKernelGGL((psd_small_reg_kernel<4, MODE>), dim3((a.count + 15) / 16), dim3(64), 0, st, a); break;
    case 1: hipLaunchKernelGGL((psd_small_reg_kernel<8, MODE>), dim3((a.count + 7) / 8), dim3(64), 0, st, a); break;
    case 2: hipLaunchKernelGGL((psd_small_reg_kernel<16, MODE>), dim3((a.count + 3) / 4), dim3(64), 0, st, a); break;
    case 3: hipLaunchKernelGGL((psd_small_reg_kernel<32, MODE>), dim3((a.count + 1) / 2), dim3(64), 0, st, a); break;
    case 4: hipLaunchKernelGGL((psd_small_reg_kernel<64, MODE>), dim3(a.count), dim3(64), 0, st, a); break;
    case 5:
      if (maxn <= 64) return launch_wg<64, MODE>(a, maxn, st);
      if (maxn <= 128) return launch_wg<128, MODE>(a, maxn, st);
      return launch_wg<256, MODE>(a, maxn, st);
    default: break;      // class 6 (blocks beyond the LDS of one workgroup): eig_large.hip, from the callers
  }
  CUADMM_HIP_TRY(hipGetLastError());
  return CUADMM_OK;
}

// Xproj = Pi_+(Xb) over all blocks of the plan (device pointers, svec layout)
// Size classes are independent: with `overlap` (engine-owned plans) every class and the sign path run on their own
// stream between a fork and a join event on `st`, so small classes (a moment relaxation has a handful of blocks per
// size) overlap instead of queueing behind each other.
// schedule hints age: one lift step fewer every 16th projection, so that a block whose spectrum got easier finds out
__global__ void hint_decay_kernel(int* hint, int n) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i < n && hint[i] > 1) hint[i] -= 1;
}

// Blocks of the one-wavefront-per-block sign kernels (classes 3 and 4) can take the iteration's vector work with them
bool PsdPlan::fusable() const {
  return eig_rank == 0 && opt.debug != 1 && opt.n32_sign && opt.mid != 1 && fused_blocks() > 0 && vec_len < 0x7fffffffLL;
}

bool PsdPlan::sort_by_steps_host(const int* steps_host, std::vector<std::pair<int, int>>& ranges) {
  // ranges served by the one-wavefront kernels: class 2 (sign16), class 3, class 4 (wave4: the n > 48 members and the others)
  ranges.clear();
  if (sign16 && cls_count[2] > 0) ranges.push_back({cls_begin[2], cls_count[2]});
  if (cls_count[3] > 0) ranges.push_back({cls_begin[3], cls_count[3]});
  if (wave4 && cls_count[4] > 0) {
    if (cls4_big > 0) ranges.push_back({cls_begin[4], cls4_big});
    if (cls_count[4] > cls4_big) ranges.push_back({cls_begin[4] + cls4_big, cls_count[4] - cls4_big});
  }
  std::vector<PsdDesc> tmp;
  for (auto& rg : ranges) {   // counting sort by steps (<= SignSched::kCap), descending, stable
    constexpr int K = SignSched::kCap + 2;
    int cnt[K + 1] = {0};
    auto key = [&](const PsdDesc& d) { const int st = steps_host[d.id]; return K - 1 - (st < 0 ? 0 : (st > K - 1 ? K - 1 : st)); };
    for (int q = 0; q < rg.second; ++q) cnt[key(h_desc[rg.first + q]) + 1]++;
    for (int k = 0; k < K; ++k) cnt[k + 1] += cnt[k];
    tmp.resize((size_t)rg.second);
    for (int q = 0; q < rg.second; ++q) tmp[(size_t)cnt[key(h_desc[rg.first + q])]++] = h_desc[rg.first + q];
    std::copy(tmp.begin(), tmp.end(), h_desc.begin() + rg.first);
  }
  return !ranges.empty();
}

bool PsdPlan::compute_dominant_geometry() const {
  double w[4] = {0, 0, 0, 0}, tot = 0;                     // NT = 1 .. 4
  for (int c = 2; c <= 4; ++c) {
    if ((c == 2 && !sign16) || (c == 4 && !wave4)) continue;
    for (int q = 0; q < cls_count[c]; ++q) {
      const int n = h_blk[h_ids[cls_begin[c] + q]];
      const int nt = n <= 16 ? 1 : (n <= 32 ? 2 : (n <= 48 ? 3 : 4));
      const double x = (double)nt * nt * nt;               // the tile is what is paid for
      w[nt - 1] += x; tot += x;
    }
  }
  for (double x : w) if (x >= 0.9 * tot && tot > 0) return true;
  return false;
}

int PsdPlan::reorder_by_steps(const int* steps_host, hipStream_t st) {
  std::vector<std::pair<int, int>> ranges;
  sort_by_steps_host(steps_host, ranges);
  for (auto& rg : ranges) {
    int rc = staged_h2d(d_desc + rg.first, &h_desc[rg.first], sizeof(PsdDesc) * (size_t)rg.second, st);
    if (rc) return rc;
  }
  return CUADMM_OK;
}

int PsdPlan::reorder_by_steps_async(const int* steps_host, hipStream_t st) {
  std::vector<std::pair<int, int>> ranges;
  if (!sort_by_steps_host(steps_host, ranges)) return CUADMM_OK;
  std::copy(h_desc.begin(), h_desc.end(), h_desc_pin);
  for (auto& rg : ranges)
    CUADMM_HIP_TRY(hipMemcpyAsync(d_desc + rg.first, h_desc_pin + rg.first, sizeof(PsdDesc) * (size_t)rg.second, hipMemcpyHostToDevice, st));
  return CUADMM_OK;
}

// partial-sum slots in launch order: class 2 (when it runs the sign kernel), 3, 4 -- as PsdPlan::project hands them out
void PsdPlan::fused_slots(std::vector<int>& slot_of) const {
  slot_of.assign((size_t)nblk, -1);
  int slot = 0;
  for (int c = 2; c <= 4; ++c) {
    if ((c == 2 && !sign16) || (c == 4 && !wave4)) continue;
    for (int q = 0; q < cls_count[c]; ++q) slot_of[h_ids[cls_begin[c] + q]] = slot++;
  }
}

// svec elements outside the fused blocks, ascending (the stand-alone vector kernels run over this list)
int PsdPlan::build_rest_index() {
  if (d_rest) { hipError_t e = hipFree(d_rest); (void)e; d_rest = nullptr; }
  std::vector<int> rest;
  long long off = 0;
  for (int k = 0; k < nblk; ++k) {
    const long long len = blk_svec_len(h_blk[k]);
    const int c = h_blk[k] > 0 && h_blk[k] < sign_min ? class_of(h_blk[k]) : -1;
    if (!(c == 3 || (c == 4 && wave4) || (c == 2 && sign16)))
      for (long long i = off; i < off + len; ++i) rest.push_back((int)i);
    off += len;
  }
  n_rest = (long long)rest.size();
  if (n_rest > 0) {
    CUADMM_HIP_TRY(hipMalloc(&d_rest, sizeof(int) * rest.size()));
    { int rc_ = staged_h2d(d_rest, rest.data(), sizeof(int) * rest.size()); if (rc_) return rc_; }
  }
  return CUADMM_OK;
}

int PsdPlan::project(const double* Xb, double* Xproj, hipStream_t st, const SignFuse* fz) const {
  if (d_hint && !(fz && fz->rec) && (++n_project & 15) == 0) {   // closed blocks age their hints themselves (psd_sign_closed.h)
    hipLaunchKernelGGL(hint_decay_kernel, dim3((nblk + 255) / 256), dim3(256), 0, st, d_hint, nblk);
    CUADMM_HIP_TRY(hipGetLastError());
  }
  const bool psd_debug = opt.debug != 0;
  const bool no_overlap = !opt.overlap || psd_debug;
  if (fz && !fusable()) { set_error("psd: fused projection requested on a plan that cannot fuse"); return CUADMM_ERR_INVALID; }
  int lanes = sign.empty() ? 0 : 1;
  for (int c = 0; c < kNumPsdClasses; ++c) lanes += cls_count[c] > 0;
  const bool fork = overlap && !no_overlap && st != nullptr && lanes > 1;
  if (fork && !ev_fork) {
    CUADMM_HIP_TRY(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    for (int i = 0; i < kNumPsdClasses; ++i) {
      CUADMM_HIP_TRY(hipStreamCreateWithFlags(&aux[i], hipStreamNonBlocking));
      CUADMM_HIP_TRY(hipEventCreateWithFlags(&ev_done[i], hipEventDisableTiming));
    }
  }
  if (fork) CUADMM_HIP_TRY(hipEventRecord(ev_fork, st));
  hipStream_t main_st = st;
  // 16 < n <= 32: option psd_n32 = 0 -> register eigensolver (psd_small_reg.h), default the one-wavefront sign kernel
  // (0.256 vs 0.92 ms per 10 000 x 32 blocks on MI355X, 0.05 vs 0.35 ms latency for a single block)
  const bool sign32 = opt.n32_sign != 0;
  const bool one_side = fork && !sign.empty();
  bool side_forked = false;
  int last_class = -1;
  for (int c = 0; c < kNumPsdClasses; ++c) if (cls_count[c] > 0) last_class = c;
  // without a sign path the class of the LARGEST blocks (the longest chain: pendulum N = 80's 80 blocks of n = 55) stays on the caller's
  // stream: it starts right behind the kernel before it instead of behind a fork event, and the join waits for the short classes only
  const int main_class = (fork && sign.empty()) ? last_class : -1;
  for (int cc = 0; cc < kNumPsdClasses; ++cc) {
    // the class that stays on the caller's stream (the longest chain) is launched FIRST, the others in ascending order
    const int c = main_class < 0 ? cc : (cc == 0 ? main_class : (cc <= main_class ? cc - 1 : cc));
    if (cls_count[c] == 0) continue;
    if (fork) {   // the sign path keeps the main stream (it is the longest chain)
      // beside a sign path (hundreds of microseconds) the small classes (tens each) share ONE side stream, one after the other: one
      // fork and one join instead of one per class (each costs ~10 us of idle queue), and none of them can land on the hardware queue
      // of the main stream (streams are multiplexed onto four queues: the n = 28 class of PlanarHand_N=1 ran in FRONT of the sign path)
      if (c == main_class) st = main_st;
      else {
        st = one_side ? aux[0] : aux[c];
        if (!one_side || !side_forked) CUADMM_HIP_TRY(hipStreamWaitEvent(st, ev_fork, 0));
        side_forked = true;
      }
    }
    PsdArgs a{};
    a.in = Xb; a.out = Xproj; a.Wout = nullptr; a.info = d_fail;
    a.ids = d_ids + cls_begin[c]; a.boff = d_off; a.bn = d_n; a.desc = d_desc + cls_begin[c];
    a.count = cls_count[c]; a.n_uniform = 0; a.steps = d_steps;
    a.eig_rank = (eig_rank > 0 && rank_active) ? eig_rank : 0;
    a.hint = d_hint;
    long long* dbg = nullptr;
    const int nwg = (cls_count[c] + 1) / 2 + 4;
    if (opt.debug == 1 && eig_rank == 0 && !fz && ((c == 2 && sign16) || (c == 3 && sign32) || (c == 4 && wave4))) {
      // phase ticks of the one-wavefront kernels (CUADMM_PSD_DEBUG=1 / option psd_debug), unfused
      std::vector<long long> h((size_t)cls_count[c] * 10, 0);
      long long* d = nullptr;
      CUADMM_HIP_TRY(hipMalloc(&d, sizeof(long long) * h.size()));
      CUADMM_HIP_TRY(hipMemset(d, 0, sizeof(long long) * h.size()));
      a.dbg = d;
      int rc2 = CUADMM_OK;
      if (c == 2) rc2 = launch_sign_wave<1, 8>(a, 0, cls_count[c], st);
      else if (c == 3) rc2 = launch_sign_wave<2, 4>(a, 0, cls_count[c], st);
      else {
        rc2 = launch_sign_wave<4, 1>(a, 0, cls4_big, st);
        if (!rc2) rc2 = launch_sign_wave<3, 2>(a, cls4_big, cls_count[c] - cls4_big, st);
      }
      if (rc2) return rc2;
      CUADMM_HIP_TRY(hipStreamSynchronize(st));
      { int rc_ = staged_d2h(h.data(), d, sizeof(long long) * h.size(), st); if (rc_) return rc_; }
      double ph[7] = {0, 0, 0, 0, 0, 0, 0};
      for (int w = 0; w < cls_count[c]; ++w) for (int q = 0; q < 7; ++q) ph[q] += (double)h[(size_t)w * 10 + q];
      const double nw = cls_count[c];
      fprintf(stderr, "[psd debug] class %d: %d blocks: ticks/block prologue %.0f (zero fill done at %.0f, first batch of loads back at %.0f, tile written at %.0f) "
                      "iteration %.0f (%.1f steps, %.0f per step) epilogue %.0f\n",
              c, cls_count[c], ph[0] / nw, ph[4] / nw, ph[5] / nw, ph[6] / nw, ph[1] / nw, ph[3] / nw, ph[1] / std::max(ph[3], 1.0), ph[2] / nw);
      { hipError_t e = hipFree(d); (void)e; }
      if (fork && c != main_class && (!one_side || c == last_class)) CUADMM_HIP_TRY(hipEventRecord(ev_done[c], st));
      continue;
    }
    if (c == 3 && fz && fz->iters > 1 && opt.debug >= 2) {   // developer aid (psd_debug = 2): phase ticks of the batched launches at full occupancy
      std::vector<long long> h((size_t)cls_count[c] * 16, 0);
      long long* d = nullptr;
      CUADMM_HIP_TRY(hipMalloc(&d, sizeof(long long) * h.size()));
      CUADMM_HIP_TRY(hipMemset(d, 0, sizeof(long long) * h.size()));
      a.dbg = d;
      SignFuse f2 = *fz;
      f2.full = range_full[1];
      int rc2 = launch_sign_wave32(a, 0, cls_count[c], st, opt, &f2);
      if (rc2) return rc2;
      CUADMM_HIP_TRY(hipStreamSynchronize(st));
      { int rc_ = staged_d2h(h.data(), d, sizeof(long long) * h.size(), st); if (rc_) return rc_; }
      double ph[16] = {0};
      for (int w = 0; w < cls_count[c]; ++w) for (int q = 0; q < 16; ++q) ph[q] += (double)h[(size_t)w * 16 + q];
      const double nw = cls_count[c];
      for (double& x : ph) x /= nw;
      // stamps 4..9 are offsets from the start of the task; ph[0..2] are the phase lengths
      const double it0 = ph[0], ep0 = ph[0] + ph[1];
      fprintf(stderr, "[cu debug] %d blocks x %d iterations: ticks/task prologue %.0f [solve done %.0f, loads back %.0f, gather done %.0f] iteration %.0f (%.2f steps, %.0f per step) "
                      "epilogue %.0f [Xb rebuilt +%.0f, P stored +%.0f, walk done +%.0f] total %.0f | outside the body: claim + flag wait %.0f, arguments + descriptor %.0f, "
                      "body call %.0f, release + flag %.0f\n",
              cls_count[c], fz->iters, ph[0], ph[4], ph[5], ph[6], ph[1], ph[3], ph[1] / std::max(ph[3], 1.0), ph[2], ph[7] - ep0, ph[8] - ep0, ph[9] - ep0,
              ph[0] + ph[1] + ph[2], ph[10], ph[11], ph[12], ph[13]);
      {
        double ck = 0, rt = 0;
        const int ng = std::min(cls_count[c], 256);
        for (int w = 0; w < ng; ++w) { ck += (double)h[(size_t)w * 16 + 14]; rt += (double)h[(size_t)w * 16 + 15]; }
        fprintf(stderr, "[cu debug] launch: %.3f ms, s_memtime runs at %.3f GHz\n", rt / ng * 1e-5, ck / std::max(rt, 1.0) * 0.1);
      }
      (void)it0;
      { hipError_t e = hipFree(d); (void)e; }
      if (fork && c != main_class && (!one_side || c == last_class)) CUADMM_HIP_TRY(hipEventRecord(ev_done[c], st));
      continue;
    }
    if (c == 3 && opt.debug == 1 && !sign32) {   // phase cycles of the register eigensolver
      CUADMM_HIP_TRY(hipMalloc(&dbg, sizeof(long long) * 8 * (size_t)nwg));
      CUADMM_HIP_TRY(hipMemset(dbg, 0, sizeof(long long) * 8 * (size_t)nwg));
      a.dbg = dbg;
    }
    int rc;
    if (c == 4 && opt.mid != 1 && eig_rank == 0) {   // members are sorted by size, largest first: [0, cls4_big) have n > 48
      // one wavefront per block (psd_sign_wave.h) from psd_wave4_min blocks on, else one workgroup per block (latency); option
      // psd_mid = 2 forces the one-workgroup kernels, 1 the register eigensolver
      if (wave4) {
        SignFuse f4, f3;
        if (fz) { f4 = *fz; f3 = *fz; f4.full = range_full[3]; f3.full = range_full[2]; }
        rc = launch_sign_wave<4, 1>(a, 0, cls4_big, st, fz ? &f4 : nullptr);
        if (!rc) rc = launch_sign_wave<3, 2>(a, cls4_big, cls_count[c] - cls4_big, st, fz ? &f3 : nullptr);
      } else {
        const bool triple = opt.lds_triple != 0 && cls_count[c] <= 256;
        rc = launch_sign_lds<64>(a, 0, cls4_big, st, triple);
        if (!rc) rc = launch_sign_lds<48>(a, cls4_big, cls_count[c] - cls4_big, st, triple);
      }
    } else if (c == 3 && sign32 && eig_rank == 0) {
      SignFuse f2;
      if (fz) { f2 = *fz; f2.full = range_full[1]; }
      rc = launch_sign_wave32(a, 0, cls_count[c], st, opt, fz ? &f2 : nullptr);
    } else if (c == 2 && sign16 && eig_rank == 0 && opt.debug != 1) {
      // 9 <= n <= 16: the same iteration on ONE 16 x 16 sub-tile, eight wavefronts per SIMD (the register eigensolver needs
      // ~20 us of dependent rotations per block; here a block is 8 MFMAs per step)
      SignFuse f1;
      if (fz) { f1 = *fz; f1.full = range_full[0]; }
      rc = launch_sign_wave<1, 8>(a, 0, cls_count[c], st, fz ? &f1 : nullptr);
    } else if (c == 6) {
      // blocks beyond one workgroup's LDS that need EIGENVALUES (the rank-limited projection; everything else of this size takes
      // the matrix-sign path): one block at a time on the whole chip (eig_large.hip)
      rc = CUADMM_OK;
      for (int q = 0; q < cls_count[c] && !rc; ++q) {
        const int k = h_ids[cls_begin[c] + q];
        rc = eig_large_project(Xb + h_off[k], Xproj + h_off[k], h_blk[k], a.eig_rank, d_fail, st, &big_ws);
      }
    } else {
      rc = launch_class<0>(c, a, cls_maxn[c], st);
    }
    if (rc) return rc;
    if (dbg) {
      std::vector<long long> h((size_t)nwg * 8);
      CUADMM_HIP_TRY(hipStreamSynchronize(st));
      { int rc_ = staged_d2h(h.data(), dbg, sizeof(long long) * h.size(), st); if (rc_) return rc_; }
      double ph[6] = {0, 0, 0, 0, 0, 0}, its = 0, slots = 0;
      long long tmin = h[0], tmax = 0;
      for (int w = 0; w < nwg; ++w) {
        for (int i = 0; i < 5; ++i) ph[i] += (double)(h[w * 8 + i + 1] - h[w * 8 + i]);
        tmin = std::min(tmin, h[w * 8]); tmax = std::max(tmax, h[w * 8 + 5]);
        its += (double)h[w * 8 + 6]; slots += (double)h[w * 8 + 7];
      }
      // MFMA utilisation of the rebuild: 2 blocks x 3 upper tiles x 8 k-steps x 64 cycles per v_mfma_f64_16x16x4
      fprintf(stderr, "[psd debug] %d waves: cycles/wave load %.0f tridiag %.0f ql %.0f handoff %.0f tail %.0f | mfma rebuild %.0f cycles/wave -> MFMA busy %.1f%%\n",
              nwg, ph[0] / nwg, ph[1] / nwg, ph[2] / nwg, ph[3] / nwg, ph[4] / nwg, its / nwg, 100.0 * (2 * 3 * 8 * 64.0) / (its / nwg));
      (void)slots; (void)tmin; (void)tmax;
      { hipError_t e = hipFree(dbg); (void)e; }
    }
    if (fork && c != main_class && (!one_side || c == last_class)) CUADMM_HIP_TRY(hipEventRecord(ev_done[c], st));
  }
  st = main_st;
  if (n_free > 0 && Xb != Xproj) {   // unconstrained blocks: Xproj = Xb on their svec ranges
    hipLaunchKernelGGL(copy_ranges_kernel, dim3(n_free), dim3(256), 0, st, Xb, Xproj, d_free_off, d_free_len);
    CUADMM_HIP_TRY(hipGetLastError());
  }
  if (!sign.empty()) {
    int rc = sign.project(Xb, Xproj, d_off, d_n, d_fail, st);
    if (rc) return rc;
  }
  if (fork)
    for (int c = 0; c < kNumPsdClasses; ++c)
      if (cls_count[c] > 0 && c != main_class && (!one_side || c == last_class)) CUADMM_HIP_TRY(hipStreamWaitEvent(st, ev_done[c], 0));
  return CUADMM_OK;
}

// pad[0] of every descriptor <- aux[block id] (the engine's closed-block header, psd_fuse.h: closed_hdr_pack); the reordering
// moves whole descriptors, so it stays with its block
int PsdPlan::set_desc_aux(const std::vector<int>& aux_of_block) {
  if ((int)aux_of_block.size() != nblk) { set_error("psd: set_desc_aux: %d values for %d blocks", (int)aux_of_block.size(), nblk); return CUADMM_ERR_INVALID; }
  if (h_desc.empty() || !d_desc) return CUADMM_OK;
  for (auto& d : h_desc) d.pad[0] = aux_of_block[(size_t)d.id];
  return staged_h2d(d_desc, h_desc.data(), sizeof(PsdDesc) * h_desc.size());
}

// entry e of the tile table of the one-wavefront geometry that serves a block of size n (psd_sign_closed.h: SwcTab<NT>)
unsigned psd_closed_tab_entry(int n, int e) {
  static const SwcTab<1> t1; static const SwcTab<2> t2; static const SwcTab<3> t3; static const SwcTab<4> t4;
  if (n <= 16) return e < SwcTab<1>::N ? t1.v[e] : 0u;
  if (n <= 32) return e < SwcTab<2>::N ? t2.v[e] : 0u;
  if (n <= 48) return e < SwcTab<3>::N ? t3.v[e] : 0u;
  return e < SwcTab<4>::N ? t4.v[e] : 0u;
}

int PsdPlan::fail_count(hipStream_t st) const {
  int h = 0;
  if (!d_fail) return 0;
  if (staged_d2h(&h, d_fail, sizeof(int), st)) return -1;      // never a runtime copy into pageable memory (staging.hip)
  return h;
}

// `count` dense n x n matrices: eigenvectors in place + ascending eigenvalues
int psd_batch_eig(double* mat, double* W, int* info, int n, int count, hipStream_t st) {
  if (n < 1 || count < 0) { set_error("batch_eig: bad n/count"); return CUADMM_ERR_INVALID; }
  if (n > kEigLargeMax) { set_error("batch_eig: n=%d > %d", n, kEigLargeMax); return CUADMM_ERR_INVALID; }
  if (n >= kEigLargeMin) {
    // one matrix at a time on the whole chip (eig_large.hip): tridiagonalisation, bisection, inverse iteration, Cholesky-QR
    for (int i = 0; i < count; ++i) {
      int rc = eig_large(mat + (size_t)i * n * n, W + (size_t)i * n, info ? info + i : nullptr, n, st);
      if (rc) return rc;
    }
    return CUADMM_OK;
  }
  if (count == 0) return CUADMM_OK;
  const int c = psd_class_of(n);
  PsdArgs a{};
  a.in = mat; a.out = mat; a.Wout = W; a.info = info; a.ids = nullptr; a.boff = nullptr; a.bn = nullptr;
  a.count = count; a.n_uniform = n;
  int rc = launch_class<1>(c, a, n, st);
  return rc;
}

}  // namespace cuadmm
