// Switches of the projection planner and its kernels.  ONE INSTANCE PER PLAN (PsdPlan::opt): two solvers in a process can
// choose differently, and every variant is reachable by a test through cuadmm_set_option (keys "psd_*").  Nothing here is
// cached process-wide; from_env() gives the defaults and reads the developer-aid variables.
#pragma once
#include <cstdlib>
#include <cstring>
#include <string>

namespace cuadmm {

struct PsdOptions {
  int n16_sign = 1;        // psd_n16: 9 <= n <= 16 on the one-wavefront sign kernel (0: register eigensolver)
  int n32_sign = 1;        // psd_n32: 17 <= n <= 32 likewise
  int mid = 0;             // psd_mid: 33 <= n <= 64: 0 = by count (psd_wave4_min), 1 = register eigensolver, 2 = one workgroup per block
  int wave4_min = 1024;    // psd_wave4_min: blocks of a 33 <= n <= 64 class from which one wavefront per block wins over one workgroup
  int w32_occ = 4;         // psd_w32_occ: wavefronts per SIMD of the n <= 32 one-wavefront kernels (3 | 4)
  int cu_occ = 4;          // psd_cu_occ: the same for the launches that run several iterations (psd_sign_closed_cu_kernel)
  int lds_triple = 1;      // psd_lds_triple: one-workgroup kernels of 33 <= n <= 64 with a third LDS matrix when the class has <= 256 blocks
  int sign_min = 65;       // psd_sign_min: blocks from this size on take the batched-GEMM matrix-sign path
  int overlap = 1;         // psd_overlap: size classes on their own streams
  // batched-GEMM path (psd_large.hip)
  int lg_tile = 0;         // psd_lg_tile: 0 = by size, 32 | 64 forces the tile
  int lg_pad32 = 1;        // psd_lg_pad32: groups on 32 x 32 tiles pad to a multiple of 32
  int lg_decide = 0;       // psd_lg_decide: 0 = by tile count, 1 = decisions inside the product kernels, 2 = a separate kernel
  int lg_merge = 1;        // psd_lg_merge: one-launch groups of different padded sizes share ONE launch (workspaces of their own)
  int lg_clean = 0;        // psd_lg_clean: 1 = groups padded to at most 512 take the CLEAN mega-lift where it pays (sign_sched.h: two step slots, no cap).  Built, parity-green and OFF: on the shipped inputs it never pays (NOTEBOOK.md "Round 6")
  int lg_fuse = 1;         // psd_lg_fuse: the one-launch kernel does its own prologue (svec -> X0, column sums, S, state) and epilogue (svec store): one launch instead of three
  int lg_cluster_wgs = 224;  // psd_lg_cluster_wgs: the largest grid the one-launch sign kernel takes (its workgroups must be co-resident: four of them fit a CU)
  int lg_cluster = 1;      // psd_lg_cluster: a handful of mid-size blocks run their whole sign iteration in one launch (per-member barriers)
  int graph = 0;           // psd_graph: replay the launch sequence from a hipGraph (measured: no gain)
  int sign_maxsteps = 0;   // psd_sign_maxsteps: cap of the schedule (0: SignSched::kCap)
  int sign_sync = 1;       // psd_sign_sync: poll "members not finished" between chunks of steps
  int sign_ws_mb = 8192;   // psd_sign_ws_mb: workspace bound of a group
  int debug = 0;           // developer aid: CUADMM_PSD_DEBUG (phase ticks of the kernels on stderr; serialises the classes)

  // the defaults: each option's environment variable (round 1 / 2 names) is consulted ONCE here, per plan -- never cached
  static PsdOptions from_env() {
    PsdOptions o;
    struct { const char* name; int* field; } tab[] = {
        {"CUADMM_PSD_DEBUG", &o.debug},           {"CUADMM_PSD_WAVE4_MIN", &o.wave4_min}, {"CUADMM_PSD_SIGN_MIN", &o.sign_min},
        {"CUADMM_PSD_W32_OCC", &o.w32_occ},       {"CUADMM_PSD_CU_OCC", &o.cu_occ},       {"CUADMM_PSD_OVERLAP", &o.overlap},
        {"CUADMM_PSD_N16", &o.n16_sign},    {"CUADMM_PSD_N32", &o.n32_sign},
        {"CUADMM_PSD_MID", &o.mid},             {"CUADMM_PSD_LG_CLUSTER", &o.lg_cluster}, {"CUADMM_PSD_LDS_TRIPLE", &o.lds_triple},
        {"CUADMM_PSD_LG_MERGE", &o.lg_merge},   {"CUADMM_PSD_LG_CLEAN", &o.lg_clean},
        {"CUADMM_PSD_LG_FUSE", &o.lg_fuse},     {"CUADMM_PSD_LG_CLUSTER_WGS", &o.lg_cluster_wgs}};
    for (auto& t : tab)
      if (const char* e = getenv(t.name)) {
        // historical spellings: N16 / N32 = "eig" (register eigensolver), MID = "eig" | "lds"
        if (!strcmp(e, "eig")) *t.field = (t.field == &o.mid) ? 1 : 0;
        else if (!strcmp(e, "lds")) *t.field = 2;
        else *t.field = atoi(e);
      }
    if (o.debug < 0) o.debug = 0;
    return o;
  }
  // returns false when the key is not one of this struct's
  bool set(const std::string& k, double value) {
    const int v = (int)value;
    if (k == "psd_n16") n16_sign = v;
    else if (k == "psd_n32") n32_sign = v;
    else if (k == "psd_mid") mid = v;
    else if (k == "psd_wave4_min") wave4_min = v;
    else if (k == "psd_w32_occ") w32_occ = v == 3 ? 3 : 4;
    else if (k == "psd_cu_occ") cu_occ = v == 3 ? 3 : 4;
    else if (k == "psd_lds_triple") lds_triple = v;
    else if (k == "psd_sign_min") sign_min = v < 65 ? 65 : v;
    else if (k == "psd_overlap") overlap = v;
    else if (k == "psd_lg_tile") lg_tile = v;
    else if (k == "psd_lg_pad32") lg_pad32 = v;
    else if (k == "psd_lg_decide") lg_decide = v;
    else if (k == "psd_lg_cluster") lg_cluster = v;
    else if (k == "psd_lg_merge") lg_merge = v;
    else if (k == "psd_lg_clean") lg_clean = v;
    else if (k == "psd_lg_fuse") lg_fuse = v;
    else if (k == "psd_lg_cluster_wgs") lg_cluster_wgs = v < 8 ? 8 : (v > 960 ? 960 : v);
    else if (k == "psd_graph") graph = v;
    else if (k == "psd_sign_maxsteps") sign_maxsteps = v;
    else if (k == "psd_sign_sync") sign_sync = v;
    else if (k == "psd_sign_ws_mb") sign_ws_mb = v < 1 ? 1 : v;
    else if (k == "psd_debug") debug = v;
    else return false;
    return true;
  }
};

}  // namespace cuadmm
