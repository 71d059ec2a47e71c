// PSD projection of blocks with n > 64 through the matrix sign function on the fp64 matrix cores (psd_large.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "psd_options.h"

namespace cuadmm {

struct ClusterMulti;

struct SignPsd {
  // pred: steps the previous projection needed.  merged: the group runs its whole sign iteration in ONE launch shared with the other
  // merged groups (lg_sign_cluster_kernel over several padded sizes), so it has a workspace of its own (ws_off: elements into X0 / S /
  // Y / T; mem_off: members into d_state / d_done / d_bar; part_off: into each half of d_part; slot: which [2] of d_group and which table
  // of d_xcc); the others run on the caller's stream one after the other in the shared region (all offsets 0)
  struct Group { int N = 0, begin = 0, count = 0, pred = 0, mem_off = 0, slot = 0; bool merged = false; size_t ws_off = 0, part_off = 0, cs_off = 0;
                 mutable int bar_par = 0; };      // cs_off: into colsum_f; bar_par: which of the two barrier-counter sets the next fused projection uses
  PsdOptions opt;                            // the owner's switches (PsdPlan::build copies its own)
  std::vector<Group> groups;                 // same padded size N, bounded workspace
  int* d_ids = nullptr;                      // block ids, group after group
  int* d_steps = nullptr;                    // not owned; when set: Newton-Schulz steps taken per block
  int* d_hint = nullptr;                     // not owned; schedule warm start per block (lift steps of the previous projection)
  int hint_max_n = 512;                      // ... for the groups padded to at most this size: mid-size blocks gain (PlanarHand_N=10's 81 blocks of
                                             // 66 ... 120: projection 1.03 -> 0.98 ms, taha1a 1.08 -> 1.02), one n = 2 000 block loses a step (C3 17 -> 18).
                                             // A property of the group, not of the path: launches and one-launch kernel stay bit-identical
  double *X0 = nullptr, *S = nullptr, *Y = nullptr, *T = nullptr, *colsum = nullptr, *scale = nullptr;
  double* Mw = nullptr;                      // fifth matrix per member: M = R - R Y of a clean mega-lift (allocated when a group padded to <= clean_max_n exists)
  double* colsum_f = nullptr;                // fused one-launch prologue: [group][member][LG_CS_ROWS][N] column-sum chunks (every group its own: merged groups run together)
  int* d_cont = nullptr;                     // [parity][member]: the next step is a clean mega-lift's second slot
  int clean_max_n = 480;                     // (below 512: a single block padded to 512 runs the super-block tile order, which has no second slot) like hint_max_n: a property of the group, so launches and one-launch kernel stay bit-identical and C3 pays nothing
  void* d_state = nullptr;                   // 2 x SignDevState per member of the largest group (adaptive schedule, sign_sched.h)
  void* d_done = nullptr;                    // SignDone per member
  double* d_part = nullptr;                  // per-tile partial sums of the schedule statistics (p1 | p2)
  size_t part_half = 0;
  int* d_group = nullptr;                    // [members not finished, largest step count] of the group in flight
  int* h_group = nullptr;                    // pinned host copy (polled between chunks of steps)
  unsigned* d_bar = nullptr;                 // per member, two sets (members_cap apart): barrier counter of the one-launch variant; a fused projection zeroes the other set
  mutable int shared_bar_par = 0;            // the counter set the next fused projection of a NON-merged group uses (they share the region at offset 0)
  size_t bar_stride = 0;                     // members_cap: distance between the two counter sets
  int* d_xcc = nullptr;                      // [member][tile]: XCD of every workgroup of the one-launch variant (run-time check)
  int build(const int* blk, const std::vector<int>& members);
  int launch_group(Group& g, const double* in, double* out, const long long* boff, const int* bn, int* d_fail, hipStream_t st, bool poll);
  void cluster_add(ClusterMulti& cm, const Group& g, int* d_fail, int max_steps) const;     // appends g to a one-launch set
  int cluster_run(ClusterMulti& cm, bool prologue, const double* in, double* out, const long long* boff, const int* bn, int* d_fail, hipStream_t st);
  void release();
  int project(const double* in, double* out, const long long* boff, const int* bn, int* d_fail, hipStream_t st);
  int project_launch(const double* in, double* out, const long long* boff, const int* bn, int* d_fail, hipStream_t st);
  bool allow_graph = false;                  // set by long-lived owners (the engine); one-shot plans launch directly
  hipGraphExec_t graph_exec = nullptr;       // captured launch sequence of project_launch for (g_in, g_out, ...)
  const double* g_in = nullptr; double* g_out = nullptr; const long long* g_boff = nullptr; const int* g_bn = nullptr; int* g_fail = nullptr;
  bool empty() const { return groups.empty(); }
  ~SignPsd() { release(); }
};

// C = alpha * A*B + beta * E, n x n row-major, A symmetric, n a multiple of 64 (E may be null)
int large_gemm_sym(int N, const double* A, const double* B, double alpha, double beta, const double* E, double* C, hipStream_t st);

}  // namespace cuadmm
