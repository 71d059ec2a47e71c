// Planner for the PSD projection: block descriptors on the device, size classes, launches.
#pragma once
#include <hip/hip_runtime.h>

#include "device_util.h"
#include <vector>

#include "eig_large.h"

#include "psd_fuse.h"
#include "psd_large.h"
#include "psd_options.h"

namespace cuadmm {

constexpr int kNumPsdClasses = 7;   // n<=4, <=8, <=16, <=32, <=64 (register kernels) | LDS workgroup | HBM workgroup
int psd_class_of(int n);
unsigned psd_closed_tab_entry(int n, int e);

// class member -> block, 32 bytes (one scalar load).  `slot`: the block's partial-sum slot in the fused iteration -- fixed at
// build time, so the sums of an iteration do not depend on the order in which the members are launched (longest block first
// reorders them; several iterations per launch hand them out dynamically): every variant leaves the same bits.
struct PsdDesc {
  long long off;   // svec offset
  int n;           // size
  int id;          // block id
  int slot;        // partial-sum slot of the fused launches (-1: not in a fused class)
  int pad[3];
};

struct PsdPlan {
  PsdOptions opt = PsdOptions::from_env();   // set before build()
  int nblk = 0;
  long long vec_len = 0;
  double sum_n3 = 0;
  long long* d_off = nullptr;  // nblk+1 svec offsets
  int* d_n = nullptr;          // block sizes
  int* d_ids = nullptr;        // block ids grouped by class
  PsdDesc* d_desc = nullptr;   // the same order, (off, n, id) per member
  int* d_fail = nullptr;       // number of blocks whose QL iteration hit its cap (cumulative)
  int* d_hint = nullptr;       // not owned; per block: lift steps the previous projection needed (sign_sched.h warm start)
  mutable unsigned n_project = 0;
  int* d_steps = nullptr;      // not owned; when set, the sign kernels record their Newton-Schulz step count per block
  mutable EigLargeWs big_ws;   // rank-limited projection of blocks beyond one workgroup's LDS (eig_large.hip)
  std::vector<long long> h_off;   // host copy of the svec offsets
  // blocks with n >= sign_min (default 65: everything beyond the register kernels) take the GEMM-only matrix-sign
  // path (psd_large.hip); the workgroup eigensolver kernels (classes 5, 6) then only serve cuadmm_op_batch_eig.
  // Option psd_sign_min moves the boundary (A/B measurements).
  int sign_min = 65;
  // f4: unconstrained blocks (negative size in blk) are copied through; eig_rank > 0 (set before build)
  // keeps only the eig_rank largest eigenvalues of every PSD block while rank_active (reference: dense_scalar.cu:51-57,
  // get_eig_rank_mask.cu:13-37) and routes every block through the eigensolver kernels
  int eig_rank = 0;
  bool rank_active = true;
  int n_free = 0;
  long long *d_free_off = nullptr, *d_free_len = nullptr;
  mutable SignPsd sign;
  bool range_full[4] = {false, false, false, false};   // every member of the NT = 1 / 2 / 3 / 4 range has n = 16 NT
  int cls4_big = 0;            // members of class 4 (32 < n <= 64) with n > 48: NP = 64 kernel, the rest NP = 48
  bool overlap = false;        // engine-owned plans: classes on their own streams (fork / join on the caller's stream)
  mutable hipEvent_t ev_fork = nullptr, ev_done[kNumPsdClasses] = {};
  mutable hipStream_t aux[kNumPsdClasses] = {};
  int cls_begin[kNumPsdClasses] = {0}, cls_count[kNumPsdClasses] = {0}, cls_maxn[kNumPsdClasses] = {0};

  // Fused iteration (SignFuse, psd_sign_wave.h): the blocks of classes 2, 3 and 4 (9 <= n <= 64, one wavefront per block) form
  // Xb from X / A^T y / C themselves and apply the S / X updates to their own svec ranges; the stand-alone vector kernels
  // then only visit the REST of the svec (d_rest: element indices outside those blocks).
  std::vector<int> h_blk;      // host copy of the block sizes
  int* d_rest = nullptr;
  long long n_rest = 0;
  bool fusable() const;
  int set_desc_aux(const std::vector<int>& aux_of_block);
  int fused_blocks() const { return (sign16 ? cls_count[2] : 0) + cls_count[3] + (wave4 ? cls_count[4] : 0); }
  // Several iterations per launch run one persistent workgroup per CU and tile geometry (psd_sign_closed_cu_kernel); workgroups
  // of different geometries cannot share a CU's LDS, so their launches would run one after the other instead of side by side
  // (measured on BASELINE configs[3], four geometries: 2.24 vs 1.74 ms per iteration): only when one geometry holds >= 90 % of
  // the fused blocks' work
  bool one_dominant_geometry() const { return dominant_geometry; }   // computed once by build(): the engine asks every iteration
  bool dominant_geometry = false;
  bool compute_dominant_geometry() const;
  // 32 < n <= 64 on the one-wavefront kernels (throughput: 1.3x the one-workgroup kernels in bulk) only when there are enough
  // blocks to fill the chip; a handful of blocks (moment relaxations) is a LATENCY problem, and there six / ten wavefronts per
  // block win (measured crossover: ~1000 blocks at n = 45 and at n = 64; option psd_wave4_min moves it)
  bool wave4 = false;
  bool sign16 = true;          // 9 <= n <= 16 on the one-wavefront sign kernel too (option psd_n16 = 0: register eigensolver)
  // n <= 8 on the one-wavefront sign kernel as well (one 16 x 16 sub-tile; set before build).  Slower than the register
  // eigensolver as a projection (sixteen 3 x 3 blocks share a wavefront there), but it lets a block-diagonal problem with tiny
  // blocks run its WHOLE iteration in the closed-block kernels: no stand-alone vector kernels, several iterations per launch.
  bool tiny_sign = false;
  int class_of(int n) const { return (tiny_sign && sign16 && n <= 16) ? 2 : psd_class_of(n); }
  int build_rest_index();
  // Longest block first: a launch ends with blocks running alone on their SIMD, and a block that needs 19 steps started last
  // keeps the chip waiting.  The step count of a block barely moves from one ADMM iteration to the next, so the engine now and
  // then re-sorts the members of the one-wavefront classes by the steps of the previous projection (descending, stable).
  int reorder_by_steps(const int* steps_host, hipStream_t st);
  // the same without draining the stream: the new order is sorted on the host while the device is busy and uploaded from a
  // page-locked copy by a copy command queued on `st` (it takes effect for the launches queued after it)
  int reorder_by_steps_async(const int* steps_host, hipStream_t st);
  PsdDesc* h_desc_pin = nullptr;
  bool sort_by_steps_host(const int* steps_host, std::vector<std::pair<int, int>>& ranges);
  std::vector<PsdDesc> h_desc;
  void fused_slots(std::vector<int>& slot_of) const;   // block -> partial-sum slot of the fused launches (-1: not fused)
  std::vector<int> h_ids;      // host copy of d_ids

  int build(const int* blk, int mat_num);
  void release();
  int project(const double* Xb, double* Xproj, hipStream_t st, const SignFuse* fz = nullptr) const;
  int fail_count(hipStream_t st) const;
  ~PsdPlan() { release(); }
};

int psd_batch_eig(double* mat, double* W, int* info, int n, int count, hipStream_t st);

}  // namespace cuadmm
