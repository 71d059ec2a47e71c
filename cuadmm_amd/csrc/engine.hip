// The SDP-ADMM iteration engine: MI355X-native replacement of SDPSolver::init / ::solve
// (reference src/solver.cu:27-342 and :355-823).
//
// Data layout (all fp64, resident in HBM for the whole solve):
//   X, S, C, Rd1, Xb, Xproj : svec vectors of this rank's contiguous block range
//   At_csr  : rows = local svec slots, columns = constraints in the factor's PERMUTED order
//   A_csr   : rows = constraints in permuted order, columns = local svec slots
//   out     : [A*X (m) | sum Rd^2 | sum C.X | A*(S-C) (m)]  -> one D2H per (half-)iteration
// The constraint-space vectors (y, b, Rp, rhs; length m) live on the host in permuted order, next
// to the host LDL^T factor of P(AA^T + eps I)P^T, so the reference's two scatter kernels per solve
// (perform_permutation, solver.cu:487,500) disappear: permutation is folded into the matrices.
// Host<->device traffic per iteration: m doubles up (y), 2m+2 doubles down (ADMM phase).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <iostream>
#include <numeric>

#include "device_util.h"
#include "psd_plan.h"
#include "tail_solve.h"
#include "lead_solve.h"
#include "vec_kernels.h"
#include "duo_group.h"

using namespace cuadmm;

namespace {

double wall_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  int alloc(size_t count) {
    release();
    n = count;
    if (count == 0) return CUADMM_OK;
    CUADMM_HIP_TRY(hipMalloc(&p, sizeof(T) * count));
    return CUADMM_OK;
  }
  int upload(const T* h, size_t count) {
    if (count == 0) return CUADMM_OK;
    return staged_h2d(p, h, sizeof(T) * count);   // never hipMemcpy on caller / vector memory (staging.hip)
  }
  int from(const std::vector<T>& h, size_t pad = 0) {   // pad: extra (zeroed) elements behind the data
    int rc = alloc(h.size() + pad);
    if (rc) return rc;
    n = h.size();
    if (pad) CUADMM_HIP_TRY(hipMemset(p + h.size(), 0, sizeof(T) * pad));
    return upload(h.data(), h.size());
  }
  void release() {
    if (p) { hipError_t e = hipFree(p); (void)e; }
    p = nullptr; n = 0;
  }
  ~DevBuf() { release(); }
};

template <class T>
struct PinnedBuf {
  T* p = nullptr;
  size_t n = 0;
  int alloc(size_t count) {
    release();
    n = count;
    if (count == 0) return CUADMM_OK;
    CUADMM_HIP_TRY(hipHostMalloc(&p, sizeof(T) * count, hipHostMallocDefault));
    return CUADMM_OK;
  }
  void release() {
    if (p) { hipError_t e = hipHostFree(p); (void)e; }
    p = nullptr; n = 0;
  }
  ~PinnedBuf() { release(); }
};

// Host loops over the m constraints: serial below kHostParMin, else kHostChunks fixed index ranges on the host pool
// (aat_ldlt.cpp).  Sums are formed per range and combined in range order: reproducible for any thread count.
constexpr int kHostParMin = 20000, kHostChunks = 32;
constexpr size_t kSvecPad = 64 * 40;       // >= 64 * U * batches of every tile geometry (psd_sign_closed.h)
template <class F>
static void host_ranges(int m, F&& body) {   // body(chunk, lo, hi)
  if (m < kHostParMin) { body(0, 0, m); return; }
  struct Ctx { F* f; int m; } ctx{&body, m};
  cuadmm_host_parallel_for(kHostChunks, [](int c, void* p) {
    Ctx* x = static_cast<Ctx*>(p);
    const long long lo = (long long)x->m * c / kHostChunks, hi = (long long)x->m * (c + 1) / kHostChunks;
    (*x->f)(c, (int)lo, (int)hi);
  }, &ctx);
}

enum KClass { K_ATY = 0, K_PSD = 1, K_POST = 2, K_SPMV = 3, K_COPY = 4, K_HOST = 5, K_COMM = 6, K_TAIL = 7, K_NUM = CUADMM_NUM_KCLASS };

// direct RCCL binding (symbols resolved at run time so that a process that already loaded an RCCL,
// e.g. through torch, shares it)
struct NcclUid { char internal[128]; };
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(NcclUid*) = nullptr;
  int (*CommInitRank)(void**, int, NcclUid, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  bool load() {
    if (AllReduce) return true;
    void* h = dlopen(nullptr, RTLD_NOW);
    if (!h || !dlsym(h, "ncclAllReduce")) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return false;
    lib = h;
    GetUniqueId = (int (*)(NcclUid*))dlsym(h, "ncclGetUniqueId");
    CommInitRank = (int (*)(void**, int, NcclUid, int))dlsym(h, "ncclCommInitRank");
    AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclAllReduce");
    CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    return GetUniqueId && CommInitRank && AllReduce;
  }
};
RcclApi g_rccl;

}  // namespace

struct cuadmm_solver {
  // options
  int device = 0, verbose = 1, rank = 0, world = 1, profile = 0, force_comm = 0, psd_steps = 0;
  // rank-limited projection (dormant in the reference: duo_solver.cu:428-438,843-850): keep only the eig_rank
  // largest eigenvalues of every PSD block from iteration eig_rank_begin_iter on, or once maxfeas < eig_rank_maxfeas
  int eig_rank = 0, eig_rank_begin_iter = 0;
  double eig_rank_maxfeas = 0.0;
  cuadmm_allreduce_fn allreduce = nullptr;
  void* allreduce_user = nullptr;
  void* rccl_comm = nullptr;

  bool initialised = false;
  hipStream_t st = nullptr;
  double t_init0 = 0, total_time = 0;

  // global dims
  int L_full = 0, m = 0, nblk_full = 0;
  // shard
  int blk_begin = 0, blk_end = 0;
  long long sv_begin = 0, sv_end = 0;
  long long L = 0;
  std::vector<int> blk_local;

  // host state (constraint space, permuted order unless noted)
  // "Owned constraints" mode of a sharded run: when every constraint touches the blocks of ONE rank only (block-diagonal
  // problems: the synthetic weak-scaling configurations), each rank keeps just its own constraints -- its own small
  // factor, its own PCIe traffic -- and the ranks exchange four scalars per iteration instead of all-reducing A*X and
  // replicating the whole solve.  world / rank are then 1 / 0 for the sharding logic, comm_world / comm_rank hold the
  // communicator, m_full / cons_local / sv_off / blk_off map back to the caller's numbering.
  bool local_mode = false;
  int comm_world = 1, comm_rank = 0, m_full = 0, blk_off = 0;
  int L_caller = 0, nblk_caller = 0;   // the caller's vec_len / mat_num (cuadmm_get_dims reports the caller's numbering)
  long long sv_off = 0;
  std::vector<int> cons_local;
  double ov_nb = 0, ov_nc = 0, ov_nb2 = 0;
  DevBuf<double> scal_d, yfull_d, b_d, normA_d;
  PinnedBuf<double> h_scal;
  double* h_scal_dev = nullptr;   // device mapping of h_scal
  // owned-constraints mode over > 1 ranks: the four scalars of the stopping test are formed on the device (rp_stats_kernel),
  // all-reduced on the stream and copied down with the result vector: one stream synchronisation per iteration, no
  // H2D -> collective -> D2H -> sync round trip of their own
  bool dev_scalars = false;
  // Device-side y-solve (forest_solve_kernel) when the elimination forest of the factor is many small trees (block-diagonal
  // A A^T: C2, C4, ros_2000 ...): y, A X, A(S-C), b stay in HBM, the host fetches only the four scalars of the stopping test.
  bool dev_solve = false;
  // Fused iteration (psd_fuse.h): the projection kernels of the 17 <= n <= 64 blocks form Xb themselves and apply the S / X
  // updates to their svec ranges; the stand-alone vector kernels visit only the rest of the svec (plan.d_rest).
  bool fuse = false;
  // ... and the constraint rows whose nonzeros all lie in one fused block are evaluated there too (SignFuse::lc_*); the
  // stand-alone SpMV then only visits the other rows (compact CSR + the constraint index of each row)
  struct LocalRows {
    bool active = false;
    int nlocal = 0, nrest = 0;
    double rest_avg = 1.0;
    DevBuf<LcDesc> desc;
    DevBuf<int> row, nzptr, e, rest_rp, rest_ci, rest_map;
    DevBuf<double> v, rest_v;
    std::vector<LcDesc> h_desc;      // host copies for the closed-block set-up
    std::vector<int> h_row, h_nzptr, h_e;
    std::vector<double> h_v;
  } lrows;
  // closed blocks (psd_fuse.h): every constraint is local to one fused block and no block has more than kClosedMaxRows of
  // them -> the blocks solve for their own multipliers and add their share of ||Rp||^2, b^T y inside the projection kernel
  struct ClosedSolve {
    bool active = false;
    DevBuf<ClosedRec> rec;         // one record per fused slot (psd_fuse.h)
    DevBuf<double> partials2, cl_out;
    bool out_dirty = true;         // a stand-alone kernel rewrote [A X | A (S - C)] by row: refresh the per-block copy first
    long long iters_done = 0;      // fused closed iterations launched so far (ages the schedule hints deterministically)
  } closed;
  DevBuf<double> quad_seg;         // segment sums of launch_reduce_quads (more than 16 384 fused blocks)
  bool stats_fused = false;        // this iteration's four scalars were formed by launch_reduce_quads
  int fused_nparts = 0;
  // Several iterations per launch (SignFuse::iters; closed blocks, ADMM phase): option "batch" = most iterations per launch
  // (0 / 1: off).  bt_p1 / bt_p2: the per-iteration partial arrays, bt_h: the four scalars of every iteration of the batch
  // (pinned, written by the reduction through its device mapping when no collective is needed), ck_*: the checkpoint a batch
  // starts from (restored when the stopping test or the tau rule fires inside it).
  // SDPDuoSolver's N-devices-from-one-process mode (duo_group.hip): this handle is rank 0 of a group; `in_group_call` marks the
  // calls the group makes on its own ranks.  option_log: every cuadmm_set_option so far (replayed on the group's children).
  void* group = nullptr;
  bool in_group_call = false;
  int duo_share_device = 0;
  int duo_exchange = -1;              // -1: device-side exchange when the ranks' devices can read each other, else host-staged
  long long duo_inject = 0;           // test hook: duo_group.hip, DuoGroup::inject
  std::vector<std::pair<std::string, double>> option_log;
  struct Batch {
    int max_iters = 64;
    bool allow_mixed = false;      // option "batch_mixed": batches although several tile geometries share the work (tests)
    DevBuf<double> p1, p2, scal_d, ck_X, ck_S, ck_y, ck_out;
    DevBuf<int> ck_hint;
    long long ck_iters_done = 0;
    PinnedBuf<double> h;
    double* h_dev = nullptr;
    long long pstride = 0;
    int cap = 0;                   // iterations the partial / scalar buffers were sized for (batch_alloc); K never exceeds it
    bool peers_agree = true;       // every rank of the communicator can batch (agreed at the start of each solve: a rank that batches
                                   // issues ONE collective of 4 K scalars per batch, a rank that does not issues one of 4 per iteration)
    int len = 0, pos = 0;          // iterations launched / consumed by the host loop
    bool have_ck = false;          // this batch started from a checkpoint (taken whenever something could invalidate it)
    double tau = 0, sig = 0;
    long long launches = 0, iters = 0, rollbacks = 0;
  } bt;
  // X, S (device) and y hold SCALED values after a solve until somebody looks at them (materialise): a second solve with
  // if_first = false then simply continues, instead of unscaling and rescaling three vectors and recomputing A X, A (S - C)
  bool pending_unscale = false;
  int lazy_unscale = 1;
  // Behavioural switches (cuadmm_set_option, before init).  The environment variables of round 1 / 2 only give the DEFAULTS, read
  // once per solver in its constructor: two solvers in a process can choose differently, and tests reach every variant.
  struct Switches {
    int fuse = 1;                 // "fuse": the vector work of 9 <= n <= 64 blocks inside their projection kernels
    int fuse_rows = 1;            // "fuse_rows": constraint rows local to one fused block evaluated there
    int fuse_solve = -1;          // "fuse_solve": closed blocks solve for their own multipliers (-1: whenever possible)
    int host_solve = 0;           // "host_solve": keep the y-solve on the host (no forest / lead / tail on the device)
    int host_scalars = 0;         // "host_scalars": owned constraints: the four scalars through the host
    int tail_k = -1;              // "tail_k": -1 cost model, 0 host-only factor, k > 0 forces the GPU tail size
    int local_constraints = 1;    // "local_constraints": owned-constraints sharding when the problem allows it
    int mapped_out = 1;           // "mapped_out": results through a mapped pinned buffer when no collective is needed
    int lpt = 1;                  // "lpt": longest block first
    int aty_post2 = 1;            // "aty_post2": the sGS second half in one pass
    int lead_stream = 0;          // "lead_stream": the leading sweeps on the streaming kernels only (no LDS-resident trees; A/B, tests)
    int tail_one_pass = 1;        // "tail_one_pass": the GPU tail applied in one pass over inv(L22) (0: two triangular GEMVs)
    int solve_next = 1;           // "solve_next": the y-solve of iteration k + 1 is enqueued before the host waits for iteration k (fetch_out)
    int tail_max_k = 32768;       // "tail_max_k": cap of the planner's GPU tail (<= 65 536: 3 x 8 K^2 bytes while it is built, 2 x 8 K^2 afterwards)
    int tail_dd = 0;              // "tail_dd": experiment (tail_solve.h)
    int tail_refine = 0;          // "tail_refine": accuracy mode of the tail (tail_solve.h)
    int tail_pivot = 1;           // "tail_pivot": the tail's dense LDL^T with diagonal pivoting (0: unpivoted, rounds 2 - 5)
    int tail_fat = 0;             // "tail_fat": the one-pass kernel on 512-thread workgroups with twice the rows in flight for K <= 10 240 (measured: slower; A/B)
    int tail_depth = 1;           // "tail_depth" / "tail_order": ring depth and row walk of the tail's one-pass kernel (tail_solve.hip)
    int tail_order = 2;
    int tail_group_pf = 0;        // "tail_group_pf": see TailSolve::group_pf (measured slower, off)
    int tail_zreg = 1;            // "tail_zreg": z in registers in that kernel where it fits
    int tail_rb = 0;              // "tail_rb": rows per barrier of that kernel (0: by size)
    int tail_prefetch = 1;        // "tail_prefetch": the tail's one-pass kernel keeps the next rows in flight across its barrier (0: rounds 3 - 5; A/B)
    int tail_shard = 1;           // "tail_shard": world > 1, replicated solve: every rank applies 1 / world of the tail's rows, the K partial
                                  // results are all-reduced (0: every rank applies the whole tail)
    int l21_device = 1;           // "l21_device": hybrid y-solve allowed (L21 on the device beside the tail when the forest is too deep; 0: host, 2: whenever the sweeps stay on the host)
    int lead_debug = 0;           // "lead_debug": statistics of the leading elimination forest on stderr at init (developer aid)
    double tail_pinv_tol = 0.0;   // "tail_pinv_tol": experiment (tail_solve.h)
    double pinv_tol = 0.0;        // "pinv_tol": the same for EVERY pivot of the device-side solve (tail, tree tops, leading sweeps): experiment (lead_solve.h)
    int lead_small_kb = 0;        // "lead_small_kb": LDS bound of the trees that share a workgroup in fours (lead_solve.h; 0 = chosen at build from 4 / 8 / 16)
    int lead_tops_refine = 0;     // "lead_tops_refine": one refinement step per direction in the dense tree tops (A/B: measured, no effect -- lead_solve.h)
    int lead_tops = -1;           // "lead_tops": dense tree tops (lead_solve.h): -1 = when the forest is too deep for the sweeps, 0 = never, L = always, cut at height L
    int debug_eig = 0;            // developer aid
  } sw;
  cuadmm_solver() {
    auto env = [](const char* n, int d) { const char* e = getenv(n); return e ? atoi(e) : d; };
    sw.fuse = env("CUADMM_FUSE", 1); sw.fuse_rows = env("CUADMM_FUSE_ROWS", 1); sw.fuse_solve = env("CUADMM_FUSE_SOLVE", -1);
    sw.host_solve = env("CUADMM_HOST_SOLVE", 0); sw.tail_k = env("CUADMM_TAIL_K", -1);
    sw.local_constraints = env("CUADMM_NO_LOCAL_CONSTRAINTS", 0) ? 0 : 1;
    sw.mapped_out = env("CUADMM_NO_MAPPED_OUT", 0) ? 0 : 1;
    sw.host_scalars = env("CUADMM_HOST_SCALARS", 0);
    opt_hint = env("CUADMM_PSD_HINT", 1);
    plan.opt = PsdOptions::from_env();
  }
  int opt_tiny_sign = 1;       // option "tiny_sign": 0 never, 1 closed candidates, 2 always
  bool closed_candidate = false;
  int duo_cpu_eig_on_gpu = 0;
  int opt_hint = 1;            // option "psd_hint": 0 off, 1 one-wavefront kernels + mid-size blocks of the batched-GEMM path, 2 all, 3 one-wavefront kernels only
  LeadSolve lead;               // ... or, with a split factor, the leading sweeps on the device around the GPU tail (lead_solve.hip)
  int forest_trees = 0;
  DevBuf<int> f_tree_ptr, f_tree_cols, f_Li;
  DevBuf<long long> f_Lp;
  DevBuf<double> f_Lx, f_D, y_best_d;
  std::vector<double> y_full;
  cuadmm_aat* fac = nullptr;
  TailSolve tail;              // dense trailing triangle of L on the GPU (tail.k == 0: whole solve on the host)
  std::vector<int> perm, perm_inv;
  std::vector<double> normA;      // original order
  std::vector<double> normA_p, b_p, y_p, Rp_p, y_best_p;
  bool y_registered = false;   // y_p.data() page-locked with hipHostRegister (its storage is never reallocated)
  double norm_borg = 1, norm_Corg = 1, bscale = 1, Cscale = 1, objscale = 1;
  double sig = 1, errRp = 0, errRd = 0, maxfeas = 0, pobj = 0, dobj = 0, relgap = 0, feasratio = 0;
  int prim_win = 0, dual_win = 0;
  double ratioconst = 1, sigmax = 1e3, sigmin = 1e-3;
  double best_KKT = 0, sgs_KKT = 0;
  bool have_best = false;
  int eig_fail_total = 0;

  // info
  int info_iter_num = 0;
  std::vector<double> info[8];

  // device
  DevBuf<int> At_rp, At_ci, A_rp, A_ci;
  AtyLongRows At_long;         // svec rows of At with more than 128 entries
  SpmvLongRows A_long;         // rows of A much longer than the average (trace / all-ones constraints)
  DevBuf<double> At_v, A_v;
  DevBuf<double> X, S, C, Rd1, Xb, Xproj, y_d, out_d, partials, X_best, S_best;
  DevBuf<int> steps_d, hint_d;
  std::vector<int> steps_h;
  long long lpt_next = 3, lpt_iters = 0;       // longest-block-first reordering: next event, iterations so far
  PinnedBuf<int> steps_pin;                    // batched launches: step counts fetched behind one batch, sorted during the next
  bool lpt_steps_ready = false;
  // Where the kernels write [A*X | sums | A*(S-C)]: the device buffer out_d when it has to be all-reduced, otherwise the
  // pinned host buffer h_out itself through its device mapping -- the results cross PCIe as the kernels produce them and
  // fetch_out is a stream synchronisation without a copy.
  double* out_w = nullptr;
  bool out_mapped = false;
  PinnedBuf<double> h_out, h_y;
  double A_avg_nnz = 1;
  PsdPlan plan;

  // profiling
  hipEvent_t ev0[K_NUM][2] = {}, ev1[K_NUM][2] = {};
  int ev_used[K_NUM] = {0};
  double prof_count[K_NUM] = {0}, prof_ms[K_NUM] = {0}, prof_bytes[K_NUM] = {0};

  ~cuadmm_solver() {
    if (y_registered) { hipError_t e = hipHostUnregister(y_p.data()); (void)e; }
    if (fac) cuadmm_aat_free(fac);
    for (int k = 0; k < K_NUM; ++k)
      for (int j = 0; j < 2; ++j) {
        if (ev0[k][j]) { hipError_t e = hipEventDestroy(ev0[k][j]); (void)e; }
        if (ev1[k][j]) { hipError_t e = hipEventDestroy(ev1[k][j]); (void)e; }
      }
    if (ev_early) { hipError_t e = hipEventDestroy(ev_early); (void)e; }
    if (st) { hipError_t e = hipStreamDestroy(st); (void)e; }
    if (rccl_comm && g_rccl.CommDestroy) { g_rccl.CommDestroy(rccl_comm); rccl_comm = nullptr; }
  }

  bool prof_on(int k) const { return profile == 1 || (profile == 2 && k == K_PSD); }
  void prof_begin(int k) {
    if (!prof_on(k) || ev_used[k] >= 2) return;
    hipError_t e = hipEventRecord(ev0[k][ev_used[k]], st); (void)e;
  }
  void prof_end(int k, double bytes) {
    if (!prof_on(k) || ev_used[k] >= 2) return;
    hipError_t e = hipEventRecord(ev1[k][ev_used[k]], st); (void)e;
    ev_used[k]++;
    prof_bytes[k] = bytes;
  }
  void prof_collect() {  // call after a stream synchronize
    if (!profile) return;
    for (int k = 0; k < K_NUM; ++k) {
      for (int j = 0; j < ev_used[k]; ++j) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev0[k][j], ev1[k][j]) == hipSuccess) { prof_ms[k] += ms; prof_count[k] += 1; }
      }
      ev_used[k] = 0;
    }
  }
  void prof_host(int k, double seconds) {
    if (profile != 1) return;
    prof_ms[k] += seconds * 1e3; prof_count[k] += 1;
  }

  int do_allreduce(double* buf, size_t count) {
    if (world <= 1 && !force_comm) return CUADMM_OK;
    return comm_allreduce(buf, count);
  }
  // all-reduce over the communicator (the sharding world, or comm_world in owned-constraints mode)
  int comm_allreduce(double* buf, size_t count) {
    prof_begin(K_COMM);
    int rc = 0;
    if (allreduce) {
      rc = allreduce(allreduce_user, buf, count, (void*)st);
    } else if (rccl_comm) {
      rc = g_rccl.AllReduce(buf, buf, count, /*ncclFloat64*/ 8, /*ncclSum*/ 0, rccl_comm, st);
    } else {
      set_error("world=%d but no all-reduce hook installed (cuadmm_set_allreduce / cuadmm_use_rccl)", local_mode ? comm_world : world);
      return CUADMM_ERR_COMM;
    }
    prof_end(K_COMM, (double)count * 8);
    if (rc) { set_error("all-reduce hook failed with code %d", rc); return CUADMM_ERR_COMM; }
    return CUADMM_OK;
  }

  // owned-constraints mode: v[0..n) <- sum over ranks (host values through a small device buffer; blocks)
  int allreduce_scalars(double* v, int n) {
    if (!local_mode || comm_world <= 1) return CUADMM_OK;
    // through the page-locked h_scal: the runtime must never register stack or caller memory (staging.hip)
    for (int q = 0; q < n; ++q) h_scal.p[q] = v[q];
    CUADMM_HIP_TRY(hipMemcpyAsync(scal_d.p, h_scal.p, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
    int rc = comm_allreduce(scal_d.p, (size_t)n);
    if (rc) return rc;
    CUADMM_HIP_TRY(hipMemcpyAsync(h_scal.p, scal_d.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, st));
    CUADMM_HIP_TRY(hipStreamSynchronize(st));
    for (int q = 0; q < n; ++q) v[q] = h_scal.p[q];
    return CUADMM_OK;
  }

  // in_fused_step: the projection launch that follows solves for y itself (closed blocks)
  int host_solve(bool in_fused_step = false) {  // y_p = (P(AA^T+eps I)P^T)^-1 rhs_p
    double t0 = wall_s();
    const double isig = 1 / sig;
    if (in_fused_step && closed.active) return CUADMM_OK;
    if (dev_solve) {   // everything it needs is in HBM (out_d: [A X | sums | A(S-C)]); y_d is the result
      prof_begin(K_TAIL);
      int rc = lead.ready ? lead.solve(out_d.p, out_d.p + (size_t)m + 2, b_d.p, isig, y_d.p, tail, st) : launch_forest_solve(forest_trees, f_tree_ptr.p, f_tree_cols.p, f_Lp.p, f_Li.p, f_Lx.p, f_D.p, out_d.p, out_d.p + (size_t)m + 2,
                                   b_d.p, isig, y_d.p, st);
      prof_end(K_TAIL, tail.k > 0 ? tail.shard_bytes : 0.0);     // bytes of inv(L22) this rank read (1 / world of 4 K^2 when the tail is sharded)
      return rc;
    }
    // the right-hand side is formed in y_p from the pinned result buffer (A(S-C) lives at h_out[m+2..) since the last
    // fetch) and solved in place: no copies of m-vectors on the critical path between two GPU phases
    const double* asmc = h_out.p + (size_t)m + 2;
    host_ranges(m, [&](int, int lo, int hi) {
      for (int i = lo; i < hi; ++i) y_p[i] = -asmc[i] + isig * Rp_p[i];   // solver.cu:478-482
    });
    int rc;
    if (tail.k == 0) {
      rc = cuadmm_aat_solve_permuted(fac, y_p.data(), y_p.data());     // solver.cu:494
    } else {   // sparse leading columns here, dense trailing triangle on the GPU
      // hybrid (lead_solve.h): L21 lives on the device with the tail, the host sweeps L11 only
      rc = lead.hybrid ? cuadmm_aat_solve_leading_forward11(fac, tail.k, y_p.data()) : cuadmm_aat_solve_leading_forward(fac, tail.k, y_p.data());
      double t1 = wall_s();
      if (!rc) rc = lead.hybrid ? lead.apply_l21(y_p.data(), y_registered, tail, st) : tail.solve(y_p.data() + (m - tail.k), st);
      double t2 = wall_s();
      prof_host(K_TAIL, t2 - t1);
      t0 += t2 - t1;
      if (!rc) rc = lead.hybrid ? cuadmm_aat_solve_leading_backward11(fac, tail.k, y_p.data(), lead.h_w) : cuadmm_aat_solve_leading_backward(fac, tail.k, y_p.data());
    }
    prof_host(K_HOST, wall_s() - t0);
    return rc;
  }

  int upload_y(bool force = false) {
    if (dev_solve && !force) return CUADMM_OK;     // y is produced on the device
    prof_begin(K_COPY);
    // y_p is registered (page-locked) at init: the copy engine reads it directly; it is not written again before the
    // next fetch_out has synchronised the stream
    const double* src = y_p.data();
    if (!y_registered) { std::memcpy(h_y.p, y_p.data(), sizeof(double) * (size_t)m); src = h_y.p; }
    CUADMM_HIP_TRY(hipMemcpyAsync(y_d.p, src, sizeof(double) * (size_t)m, hipMemcpyHostToDevice, st));
    prof_end(K_COPY, (double)m * 8);
    return CUADMM_OK;
  }

  // D2H of out_d[first, first+count) after the optional all-reduce; blocks until it has landed.
  // solve_next (round 5): the y-solve of the NEXT iteration is enqueued behind this iteration's last kernel BEFORE the host waits --
  // and the host then waits for an event recorded in front of it, not for the stream.  The reference solves for y at the top of every
  // pass through the loop, the breaking one included (solver.cu:478-500 precede the break), so that solve is never speculative; its only
  // host input is sigma, which the caller knows will not change in this iteration's step 5.  The GPU runs the solve (0.2 - 0.4 ms on the
  // moment relaxations) while the host goes through the stopping test and enqueues the rest of the next iteration: the idle gap at
  // the iteration boundary (kernel trace: the largest one of a latency-bound iteration) is gone.
  bool y_early = false;           // y_d already holds the y-solve of the coming iteration
  hipEvent_t ev_early = nullptr;
  int fetch_out(size_t first, size_t count, bool solve_next = false) {
    if (!out_mapped) {
      int rc = do_allreduce(out_d.p + first, count);
      if (rc) return rc;
      if ((dev_scalars || dev_solve) && first == 0) {     // [||Rp||^2, b.y, sum Rd^2, <C,X>] (of this rank -> sum over ranks), on the stream
        // without a collective in between the final stage writes the four scalars straight into the pinned host buffer
        // through its device mapping: the fetch is then a stream synchronisation without a copy command
        double* dst = (!dev_scalars && h_scal_dev) ? h_scal_dev : scal_d.p;
        if (stats_fused) stats_fused = false;               // formed by launch_reduce_quads in this iteration's fused step
        else if ((rc = launch_rp_stats(m, out_d.p, b_d.p, normA_d.p, y_d.p, bscale, out_d.p + (size_t)m, scal_d.p + 8, dst, st))) return rc;
        if (dev_scalars) {
          if ((rc = comm_allreduce(scal_d.p, 4))) return rc;
        }
        if (dst == scal_d.p) CUADMM_HIP_TRY(hipMemcpyAsync(h_scal.p, scal_d.p, sizeof(double) * 4, hipMemcpyDeviceToHost, st));
      }
      if (dev_solve) {
        if (first != 0) return CUADMM_OK;                 // sGS half step: A(S-C) is consumed on the device, nothing to wait for
      } else {
        CUADMM_HIP_TRY(hipMemcpyAsync(h_out.p + first, out_d.p + first, sizeof(double) * count, hipMemcpyDeviceToHost, st));
      }
    }
    // (measured and rejected, round 5: a one-thread kernel storing a sequence number into a mapped pinned word with the host spinning on
    // it instead of this call -- c5 2 494 -> 2 476, c1 1 290 -> 1 274 iters/s: the runtime's wait already spins, the extra launch costs)
    if (solve_next && dev_solve && !out_mapped && st != nullptr) {
      // system-scope release: the host reads the pinned h_scal / h_out behind this wait (the stream synchronisation it replaces was one)
      if (!ev_early) CUADMM_HIP_TRY(hipEventCreateWithFlags(&ev_early, hipEventDisableTiming | hipEventReleaseToSystem));
      CUADMM_HIP_TRY(hipEventRecord(ev_early, st));
      { int rc = host_solve(); if (rc) { (void)hipStreamSynchronize(st); return rc; } }     // nothing of this iteration stays in flight behind an error
      y_early = true;
      CUADMM_HIP_TRY(hipEventSynchronize(ev_early));
    } else {
      CUADMM_HIP_TRY(hipStreamSynchronize(st));
    }
    prof_collect();
    return CUADMM_OK;
  }

  int launch_aty(bool write_xb) {
    prof_begin(K_ATY);
    int rc = launch_aty_xb(write_xb, L, At_rp.p, At_ci.p, At_v.p, y_d.p, C.p, X.p, sig, Rd1.p, Xb.p, st, &At_long);
    prof_end(K_ATY, (write_xb ? 32.0 : 16.0) * (double)L + 4.0 * (double)L);
    return rc;
  }
  // after_fused: the local rows were written by the fused projection kernels of this step -- only the other rows are left
  int launch_spmv(bool doX, bool doS, bool after_fused = false) {
    if (!(after_fused && lrows.active && lrows.nrest == 0)) closed.out_dirty = true;
    if (after_fused && lrows.active) {
      if (lrows.nrest == 0) return CUADMM_OK;
      prof_begin(K_SPMV);
      int rc = launch_spmv_rows(lrows.nrest, lrows.rest_avg, lrows.rest_rp.p, lrows.rest_ci.p, lrows.rest_v.p, X.p, S.p, C.p, doX ? out_w : nullptr,
                                doS ? out_w + m + 2 : nullptr, st, nullptr, lrows.rest_map.p);
      prof_end(K_SPMV, 12.0 * (double)lrows.rest_v.n + 8.0 * lrows.nrest * ((doX ? 1 : 0) + (doS ? 1 : 0)));
      return rc;
    }
    prof_begin(K_SPMV);
    int rc = launch_spmv_rows(m, A_avg_nnz, A_rp.p, A_ci.p, A_v.p, X.p, S.p, C.p, doX ? out_w : nullptr,
                              doS ? out_w + m + 2 : nullptr, st, &A_long);
    prof_end(K_SPMV, 12.0 * (double)A_v.n + 8.0 * m * ((doX ? 1 : 0) + (doS ? 1 : 0)));
    return rc;
  }
  int launch_project() {
    prof_begin(K_PSD);
    int rc = plan.project(Xb.p, Xproj.p, st);
    prof_end(K_PSD, 16.0 * (double)L);
    return rc;
  }
  // Step 2 of a fused iteration: Rd1 / Xb on the rest of the svec, then the projection whose fused blocks also do the
  // post step `mode` (0: S, X, sums; 1: S only) on their own ranges, then the post step on the rest (+ the sums).
  int launch_fused_step(int mode, double tau, int iters = 1) {
    int rc;
    prof_begin(K_ATY);
    rc = launch_aty_xb_idx(plan.n_rest, plan.d_rest, At_rp.p, At_ci.p, At_v.p, y_d.p, C.p, X.p, sig, Rd1.p, Xb.p, st);
    prof_end(K_ATY, 36.0 * (double)plan.n_rest);
    if (rc) return rc;
    SignFuse fz{};                                     // every optional part (local rows, closed blocks) off
    fz.rp = At_rp.p; fz.ci = At_ci.p; fz.av = At_v.p; fz.y = y_d.p; fz.C = C.p;
    fz.X = X.p; fz.Rd1 = Rd1.p; fz.S = S.p; fz.partials = partials.p;
    fz.sig = sig; fz.inv_sig = 1 / sig; fz.tau_sig = tau * sig; fz.mode = mode;
    if (lrows.active) {
      fz.lc = lrows.desc.p; fz.lc_row = lrows.row.p; fz.lc_nzptr = lrows.nzptr.p; fz.lc_e = lrows.e.p; fz.lc_v = lrows.v.p;
      fz.outX = mode == 0 ? out_w : nullptr;
      fz.outS = out_w + m + 2;
    }
    if (closed.active) {
      fz.rec = closed.rec.p; fz.cl_out = closed.cl_out.p; fz.y_out = y_d.p;
      fz.iter0 = (int)(closed.iters_done & 0x3fffffff);
      closed.iters_done += iters > 1 ? iters : 1;
      fz.partials2 = closed.partials2.p;
      fz.isig = 1 / sig; fz.bscale = bscale;
      if (closed.out_dirty) {
        if ((rc = launch_closed_gather_out(closed.rec.p, plan.fused_blocks(), out_d.p, out_d.p + (size_t)m + 2, closed.cl_out.p, st))) return rc;
        closed.out_dirty = false;
      }
    }
    if (iters > 1) {   // batch: every block is closed and fused (can_batch), nothing runs beside the projection launches
      fz.iters = iters; fz.pstride = bt.pstride; fz.partials = bt.p1.p; fz.partials2 = bt.p2.p;
      prof_begin(K_PSD);
      rc = plan.project(Xb.p, Xproj.p, st, &fz);
      prof_end(K_PSD, 68.0 * (double)L * iters);
      if (rc) return rc;
      prof_begin(K_POST);
      double* dst = (!dev_scalars && bt.h_dev) ? bt.h_dev : bt.scal_d.p;
      rc = launch_reduce_quads_batch(bt.p1.p, bt.p2.p, plan.fused_blocks(), bt.pstride, iters, dst, st);
      prof_end(K_POST, 0.0);
      if (rc) return rc;
      if (dev_scalars && (rc = comm_allreduce(bt.scal_d.p, 4 * (size_t)iters))) return rc;
      if (dst == bt.scal_d.p) CUADMM_HIP_TRY(hipMemcpyAsync(bt.h.p, bt.scal_d.p, sizeof(double) * 4 * (size_t)iters, hipMemcpyDeviceToHost, st));
      bt.launches++; bt.iters += iters;
      return CUADMM_OK;
    }
    prof_begin(K_PSD);
    rc = plan.project(Xb.p, Xproj.p, st, &fz);
    prof_end(K_PSD, 68.0 * (double)(L - plan.n_rest) + 16.0 * (double)plan.n_rest);
    if (rc) return rc;
    prof_begin(K_POST);
    stats_fused = closed.active && mode == 0;
    int nparts = 0;
    rc = launch_post_rest(mode, plan.n_rest, plan.d_rest, plan.fused_blocks(), Xproj.p, Rd1.p, C.p, X.p, S.p, 1 / sig, tau * sig, partials.p,
                          out_w + (size_t)m, st, stats_fused ? &nparts : nullptr);
    if (!rc && stats_fused) {   // all four scalars of the stopping test in one launch (rp_stats is not needed this iteration)
      double* dst = (!dev_scalars && h_scal_dev) ? h_scal_dev : scal_d.p;
      if (!quad_seg.p && reduce_quads_segments(nparts, plan.fused_blocks()) > 1 && (rc = quad_seg.alloc(4 * (size_t)reduce_quads_segments(nparts, plan.fused_blocks()) + 4))) return rc;
      rc = launch_reduce_quads(partials.p, nparts, closed.partials2.p, plan.fused_blocks(), dst, out_w + (size_t)m, quad_seg.p, st);
    }
    prof_end(K_POST, (mode == 0 ? 48.0 : 32.0) * (double)plan.n_rest);
    return rc;
  }
  // --- several iterations per launch ---------------------------------------------------------------------------------
  bool can_batch() const {
    return can_batch_local() && bt.peers_agree;
  }
  bool can_batch_local() const {
    return bt.max_iters >= 2 && fuse && closed.active && dev_solve && !lead.ready && plan.n_rest == 0 && eig_rank == 0 && !out_mapped &&
           plan.fused_blocks() > 0 && (bt.allow_mixed || plan.one_dominant_geometry());
  }
  // Ranks must take the batching decision TOGETHER: it decides which collective a rank issues (shards of a heterogeneous problem
  // can differ: no dominant geometry on one, a block with more than 8 rows on another, no blocks at all on a third).  One small
  // all-reduce at the start of every solve (options may have changed since the last one).
  int batch_agree() {
    bt.peers_agree = true;
    const int cw = local_mode ? comm_world : world;
    if (cw <= 1 && !force_comm) return CUADMM_OK;
    if (!allreduce && !rccl_comm) return CUADMM_OK;      // no transport yet: the first collective of the solve reports it
    h_scal.p[0] = can_batch_local() ? 1.0 : 0.0;
    h_scal.p[1] = 1.0;
    CUADMM_HIP_TRY(hipMemcpyAsync(scal_d.p, h_scal.p, sizeof(double) * 2, hipMemcpyHostToDevice, st));
    int rc = comm_allreduce(scal_d.p, 2);
    if (rc) return rc;
    CUADMM_HIP_TRY(hipMemcpyAsync(h_scal.p, scal_d.p, sizeof(double) * 2, hipMemcpyDeviceToHost, st));
    CUADMM_HIP_TRY(hipStreamSynchronize(st));
    bt.peers_agree = h_scal.p[0] >= h_scal.p[1] - 0.5;    // every rank said yes
    return CUADMM_OK;
  }
  int batch_alloc() {
    // sized for the CURRENT option value: "batch" may be raised between solves, and a launch of K iterations writes K partial
    // arrays and 4 K scalars
    if (bt.p1.p && bt.cap >= bt.max_iters) return CUADMM_OK;
    bt.pstride = 2 * (long long)plan.fused_blocks() + 2;
    bt.cap = 0; bt.h_dev = nullptr;
    int rc;
    if ((rc = bt.p1.alloc((size_t)bt.pstride * bt.max_iters)) || (rc = bt.p2.alloc((size_t)bt.pstride * bt.max_iters)) ||
        (rc = bt.scal_d.alloc(4 * (size_t)bt.max_iters)) || (rc = bt.h.alloc(4 * (size_t)bt.max_iters)))
      return rc;
    if (!bt.ck_X.p && ((rc = bt.ck_X.alloc(L)) || (rc = bt.ck_S.alloc(L)) || (rc = bt.ck_y.alloc(std::max(m, 1))) || (rc = bt.ck_out.alloc(2 * (size_t)m + 2))))
      return rc;
    bt.cap = bt.max_iters;
    void* dp = nullptr;
    if (sw.mapped_out && hipHostGetDevicePointer(&dp, bt.h.p, 0) == hipSuccess && dp) bt.h_dev = static_cast<double*>(dp);
    else { hipError_t e = hipGetLastError(); (void)e; }
    return CUADMM_OK;
  }
  int batch_copy(bool save) {   // checkpoint of everything an iteration reads and writes: X, S, y, [A X | sums | A (S - C)]
    prof_begin(K_COPY);
    struct { double* live; double* ck; size_t n; } v[4] = {{X.p, bt.ck_X.p, (size_t)L}, {S.p, bt.ck_S.p, (size_t)L}, {y_d.p, bt.ck_y.p, (size_t)m},
                                                           {out_d.p, bt.ck_out.p, 2 * (size_t)m + 2}};
    CopyJobs jobs{};                       // ONE launch for the whole checkpoint (five copies of ~0.03 ms each were 1.3 % of a batch of 44)
    for (auto& q : v) {
      jobs.dst[jobs.count] = save ? q.ck : q.live; jobs.src[jobs.count] = save ? q.live : q.ck; jobs.nbytes[jobs.count] = (long long)(q.n * sizeof(double));
      ++jobs.count;
    }
    if (hint_d.p) {   // the schedule hints are part of the state an iteration reads and writes
      if (!bt.ck_hint.p) { int rc = bt.ck_hint.alloc(hint_d.n); if (rc) return rc; }
      jobs.dst[jobs.count] = save ? bt.ck_hint.p : hint_d.p; jobs.src[jobs.count] = save ? hint_d.p : bt.ck_hint.p; jobs.nbytes[jobs.count] = (long long)(sizeof(int) * hint_d.n);
      ++jobs.count;
    }
    { int rc = launch_copy_multi(jobs, st); if (rc) return rc; }
    if (save) bt.ck_iters_done = closed.iters_done; else closed.iters_done = bt.ck_iters_done;
    prof_end(K_COPY, 16.0 * (2.0 * (double)L + 3.0 * m + 2));
    return CUADMM_OK;
  }
  // the device ran bt.len iterations, the host schedule accepts only the first `keep` of them: back to the checkpoint and
  // forward again by `keep` iterations (bit-identical: same kernels, same inputs)
  int batch_rollback(int keep) {
    if (!bt.have_ck) { set_error("internal: batch invalidated without a checkpoint"); return CUADMM_ERR_INVALID; }
    int rc = batch_copy(false);
    if (rc) return rc;
    closed.out_dirty = true;
    const double sig_now = sig;
    sig = bt.sig;
    if (keep == 1) rc = launch_fused_step(0, bt.tau);
    else if (keep > 1) rc = launch_fused_step(0, bt.tau, keep);
    sig = sig_now;
    if (rc) return rc;
    CUADMM_HIP_TRY(hipStreamSynchronize(st));
    prof_collect();
    stats_fused = false;
    bt.len = bt.pos = 0;
    bt.rollbacks++;
    return CUADMM_OK;
  }
  // X, S, y back in the caller's units (solver.cu:814-816); a no-op unless a solve left them scaled
  int materialise();

  int launch_post_mode(int mode, double tau) {
    prof_begin(K_POST);
    int rc = launch_post(mode, L, Xproj.p, Rd1.p, C.p, X.p, S.p, 1 / sig, tau * sig, partials.p, out_w + (size_t)m, st);
    prof_end(K_POST, (mode == 0 ? 48.0 : (mode == 1 ? 32.0 : 40.0)) * (double)L);
    return rc;
  }
};

using Solver = cuadmm_solver;

// keeps the calling thread's current device across an entry point that visits several devices (the in-process group's leader):
// a caller that shares the thread with another HIP user (torch) finds its device as it left it
struct DeviceGuard {
  int dev = -1;
  DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) { dev = -1; (void)hipGetLastError(); } }
  ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

static int check_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    set_error("no HIP device available (hipGetDeviceCount: %s); this engine has no CPU fallback", hipGetErrorString(e));
    return CUADMM_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) { set_error("device %d out of range (0..%d)", device, n - 1); return CUADMM_ERR_NO_DEVICE; }
  CUADMM_HIP_TRY(hipSetDevice(device));
  return CUADMM_OK;
}

// ------------------------------------------------------------------------------------------
// SDPSolver::init, in stages.  InitIn: the caller's arrays (solver.h:208-223); InitCtx: what one stage leaves for the next.
// ------------------------------------------------------------------------------------------
namespace {
struct InitIn {
  int vec_len, con_num, At_nnz, b_nnz, C_nnz, mat_num;
  const int *At_cp, *At_ri, *b_idx, *C_idx, *blk;
  const double *At_vx, *b_vals, *C_vals, *X0, *y0, *S0;
  double sig;
};
struct InitCtx {
  std::vector<double> vals;      // A^T values with the columns normalised (get_normA)
  std::vector<int> rp, rci;      // A^T in CSR over the svec rows: row pointers, constraint of every entry
  std::vector<double> rv;
};
#define CUADMM_INIT_STAGE_PROLOGUE \
  const int vec_len = in.vec_len, con_num = in.con_num, At_nnz = in.At_nnz, b_nnz = in.b_nnz, C_nnz = in.C_nnz, mat_num = in.mat_num; \
  const int *At_cp = in.At_cp, *At_ri = in.At_ri, *b_idx = in.b_idx, *C_idx = in.C_idx, *blk = in.blk; \
  const double *At_vx = in.At_vx, *b_vals = in.b_vals, *C_vals = in.C_vals, *X0 = in.X0, *y0 = in.y0, *S0 = in.S0; \
  const double sig = in.sig; \
  const int m = con_num; \
  const long long Lf = vec_len, L = s->L; \
  std::vector<double>&vals = c.vals, &rv = c.rv; \
  std::vector<int>&rp = c.rp, &rci = c.rci; \
  int rc = CUADMM_OK; \
  (void)vec_len; (void)At_nnz; (void)b_nnz; (void)C_nnz; (void)mat_num; (void)At_cp; (void)At_ri; (void)b_idx; (void)C_idx; (void)blk; (void)At_vx; \
  (void)b_vals; (void)C_vals; (void)X0; (void)y0; (void)S0; (void)sig; (void)m; (void)Lf; (void)L; (void)vals; (void)rv; (void)rp; (void)rci; \
  (void)0;

// stage: stream, events, global dimensions
static int init_device(Solver* s, const InitIn& in, InitCtx& c) {
  CUADMM_INIT_STAGE_PROLOGUE
  if ((rc = check_device(s->device))) return rc;
  s->t_init0 = wall_s();                                  // the reference's timer starts in init (solver.cu:41-44)
  CUADMM_HIP_TRY(hipStreamCreateWithFlags(&s->st, hipStreamNonBlocking));
  if (s->profile)
    for (int k = 0; k < K_NUM; ++k)
      for (int j = 0; j < 2; ++j) { CUADMM_HIP_TRY(hipEventCreate(&s->ev0[k][j])); CUADMM_HIP_TRY(hipEventCreate(&s->ev1[k][j])); }

  s->m = m; s->L_full = vec_len; s->nblk_full = mat_num; s->sig = sig;

  return rc;
}

// stage: get_normA and A^T in CSR over the svec rows (what the factor and the device matrices are built from)
static int init_normalise(Solver* s, const InitIn& in, InitCtx& c) {
  CUADMM_INIT_STAGE_PROLOGUE
  // --- get_normA (sparse_matrix_norm.cu:11-31): norm_j = max(1,||col j||), column scaled in place
  vals.assign(At_vx, At_vx + At_nnz);
  s->normA.assign(m, 1.0);
  for (int j = 0; j < m; ++j) {
    double nrm = 0.0;
    for (int p = At_cp[j]; p < At_cp[j + 1]; ++p) nrm += vals[p] * vals[p];
    nrm = std::max(1.0, std::sqrt(nrm));
    s->normA[j] = nrm;
    for (int p = At_cp[j]; p < At_cp[j + 1]; ++p) vals[p] /= nrm;
  }

  // --- At in CSR over the svec rows (== CSC of A), solver.cu:83-88
  rp.assign((size_t)Lf + 1, 0); rci.assign((size_t)At_nnz, 0);
  rv.assign((size_t)At_nnz, 0.0);
  for (int p = 0; p < At_nnz; ++p) rp[(size_t)At_ri[p] + 1]++;
  for (long long i = 0; i < Lf; ++i) rp[i + 1] += rp[i];
  {
    std::vector<int> pos(rp.begin(), rp.end() - 1);
    for (int j = 0; j < m; ++j)
      for (int p = At_cp[j]; p < At_cp[j + 1]; ++p) {
        int q = pos[At_ri[p]]++;
        rci[q] = j; rv[q] = vals[p];
      }
  }

  return rc;
}

// stage: A A^T + eps I: ordering, host factor, GPU tail of the Schur complement with its probe solve, permutation
static int init_factor(Solver* s, const InitIn& in, InitCtx& c) {
  CUADMM_INIT_STAGE_PROLOGUE
  // --- factor of A A^T + 1e-15 I (solver.cu:91-96, cholesky_cpu.h:62-141): ordering, symbolic analysis and the sparse
  // leading columns on the host; when the cost model finds a dense tail, its Schur complement is factored (dense
  // LDL^T) and inverted on the GPU and applied as two GEMVs per solve (tail_solve.hip).
  // CUADMM_TAIL_K: 0 = everything on the host, k > 0 forces the tail size (A/B measurements).
  {
    int max_k = std::max(64, std::min(s->sw.tail_max_k, 65536));
    if (s->sw.tail_k >= 0) max_k = -std::min(std::min(s->sw.tail_k, m), 65536);
    double t0 = wall_s();
    if (max_k == 0) rc = cuadmm_aat_create(m, vec_len, rp.data(), rci.data(), rv.data(), 1e-15, &s->fac);
    else {
      // the planner may choose a (smaller) tail for the solve with dense tree tops unless the options rule that solve out
      cuadmm_aat_plan_allow_tops(s->sw.lead_tops != 0 && !s->sw.host_solve && s->sw.l21_device != 2);
      rc = cuadmm_aat_create_split(m, vec_len, rp.data(), rci.data(), rv.data(), 1e-15, max_k, &s->fac);
      cuadmm_aat_plan_allow_tops(1);
    }
    if (rc) return rc;
    double t1 = wall_s();
    int tk = cuadmm_aat_tail_k(s->fac);
    if (tk > 0) {
      const int64_t* srp; const int* sci; const double* sv;
      rc = cuadmm_aat_tail_schur(s->fac, &srp, &sci, &sv);
      s->tail.one_pass = s->sw.tail_one_pass != 0;
      s->tail.prefetch = s->sw.tail_prefetch != 0;
      s->tail.depth = s->sw.tail_depth;
      s->tail.order = s->sw.tail_order;
      s->tail.rows_per_group = s->sw.tail_rb;
      s->tail.zreg = s->sw.tail_zreg != 0;
      s->tail.group_pf = s->sw.tail_group_pf != 0;
      s->tail.fat = s->sw.tail_fat != 0;
      s->tail.dd_dot = s->sw.tail_dd != 0;
      s->tail.refine = s->sw.tail_refine != 0;
      s->tail.pivot = s->sw.tail_pivot != 0;
      s->tail.pinv_tol = std::max(s->sw.tail_pinv_tol, s->sw.pinv_tol);
      if (!rc) rc = s->tail.build_from_schur(reinterpret_cast<const long long*>(srp), sci, sv, tk, s->st);
      // The tail is applied as an explicit inverse built without pivoting (tail_solve.hip); with (nearly) dependent
      // constraints the pivots approach the regularisation 1e-15 and inv(L22) could lose accuracy silently.  Probe it with
      // a consistent right-hand side z = S x (S = the Schur complement the tail factors, lower triangle with diagonal):
      // solve on the GPU and check the BACKWARD error ||S x^ - z|| / ||z|| -- the forward error is meaningless here (moment
      // relaxations have numerically singular S: no solver recovers x, and the ADMM iteration only needs a small residual).
      // On failure fall back to the host-only factor (CHOLMOD-style substitution, no explicit inverse).
      bool tail_ok = rc == CUADMM_OK;
      if (rc == CUADMM_OK) {
        std::vector<double> x((size_t)tk), z((size_t)tk, 0.0), r((size_t)tk, 0.0);
        unsigned long long seed = 0x9e3779b97f4a7c15ull;
        for (int i = 0; i < tk; ++i) { seed = seed * 6364136223846793005ull + 1442695040888963407ull; x[i] = 0.5 + (double)(seed >> 11) * (1.0 / 9007199254740992.0); }
        auto apply_S = [&](const std::vector<double>& in, std::vector<double>& out) {
          std::fill(out.begin(), out.end(), 0.0);
          for (int i = 0; i < tk; ++i)
            for (int64_t q = srp[i]; q < srp[i + 1]; ++q) {
              const int j = sci[q];
              out[i] += sv[q] * in[j];
              if (j != i) out[j] += sv[q] * in[i];
            }
        };
        apply_S(x, z);
        std::vector<double> xh = z;
        rc = s->tail.solve(xh.data(), s->st);
        apply_S(xh, r);
        double err = 0, zn = 0;
        for (int i = 0; i < tk; ++i) { err = std::max(err, std::fabs(r[i] - z[i])); zn = std::max(zn, std::fabs(z[i])); }
        tail_ok = rc == CUADMM_OK && err <= 1e-6 * zn;
        if (!tail_ok && rc == CUADMM_OK && s->verbose)
          printf("\n A*A^T factor: the GPU tail of size %d fails its probe solve (backward error %.1e): falling back to the host-only factor\n", tk, err / std::max(zn, 1e-300));
      }
      cuadmm_aat_tail_schur_release(s->fac);
      if (rc && rc != CUADMM_ERR_FACTOR) return rc;
      if (!tail_ok) {
        s->tail.release();
        cuadmm_aat_free(s->fac);
        s->fac = nullptr;
        if ((rc = cuadmm_aat_create(m, vec_len, rp.data(), rci.data(), rv.data(), 1e-15, &s->fac))) return rc;
      }
    }
    tk = s->tail.k;
    // a rank of a sharded engine keeps only the rows of inv(L22) it applies (TailSolve::keep_shard; the reference splits its buffers over the
    // devices too, src/duo_solver.cu:269-295): 1 / world of the triangle per rank instead of W and W^T whole
    if (tk > 0 && s->world > 1 && !s->local_mode && s->sw.tail_shard != 0 && !s->sw.tail_refine && (rc = s->tail.keep_shard(s->rank, s->world, s->st))) return rc;
    if (s->verbose) {
      const long long lnz = (long long)cuadmm_aat_factor_nnz(s->fac);
      if (tk > 0)
        printf("\n A*A^T factor: nnz(L) = %lld; host part %.2fs; last %d of %d columns (%.1f%% of nnz(L)) factored and inverted on the GPU in %.2fs\n",
               lnz, t1 - t0, tk, m, 100.0 * (double)(lnz - cuadmm_aat_factor_colptr(s->fac)[m - tk]) / (double)std::max<long long>(1, lnz), s->tail.build_s);
      else
        printf("\n A*A^T factor: nnz(L) = %lld on the host in %.2fs\n", lnz, t1 - t0);
    }
  }
  s->perm.assign(cuadmm_aat_perm(s->fac), cuadmm_aat_perm(s->fac) + m);
  s->perm_inv.assign(m, 0);
  for (int i = 0; i < m; ++i) s->perm_inv[s->perm[i]] = i;

  return rc;
}

// stage: census, this rank's block range, the projection plan (closed-block candidate, schedule hints, step counts)
static int init_plan(Solver* s, const InitIn& in, InitCtx& c) {
  CUADMM_INIT_STAGE_PROLOGUE
  // --- census (analyze_blk.cu:63-99, matrix_sizes.cu:75-113)
  if (s->verbose) {
    std::vector<int> sizes, nums;
    std::vector<int> psd_blk;
    for (int k = 0; k < mat_num; ++k) if (blk[k] > 0) psd_blk.push_back(blk[k]);
    analyze_blk(psd_blk.data(), (int)psd_blk.size(), sizes, nums);
    print_blk_census(sizes, nums);
    MatrixSizes ms;
    ms.init(sizes, nums);
    ms.print();
  }

  // --- shard: contiguous block range of this rank
  std::vector<int> first;
  partition_blocks(blk, mat_num, s->world, first);
  s->blk_begin = first[s->rank]; s->blk_end = first[s->rank + 1];
  long long off = 0;
  s->sv_begin = 0;
  for (int k = 0; k < mat_num; ++k) {
    if (k == s->blk_begin) s->sv_begin = off;
    off += blk_svec_len(blk[k]);
    if (k + 1 == s->blk_end) s->sv_end = off;
  }
  if (s->blk_begin == s->blk_end) { s->sv_begin = s->sv_end = (s->blk_begin == mat_num ? off : s->sv_begin); }
  s->L = s->sv_end - s->sv_begin;
  s->blk_local.assign(blk + s->blk_begin, blk + s->blk_end);
  s->plan.eig_rank = s->eig_rank > 0 ? s->eig_rank : 0;
  {
    // Closed-block candidate: every constraint lives in ONE block of this rank, no block has more than kClosedMaxRows of them or
    // more than kFuseRowsMax nonzeros, every block fits a one-wavefront kernel (n <= 64).  Then the tiny blocks go through the sign
    // kernel too (PsdPlan::tiny_sign), so that the whole iteration of every block can run in psd_sign_closed.h.
    const auto& bl = s->blk_local;
    bool cand = s->opt_tiny_sign != 0 && s->eig_rank == 0 && !bl.empty() && m > 0 && s->world == 1;
    bool any_tiny = false;
    for (size_t k = 0; k < bl.size() && cand; ++k) { cand = bl[k] > 0 && bl[k] <= 64; any_tiny = any_tiny || bl[k] <= 8; }
    if (cand && any_tiny && s->opt_tiny_sign == 1) {
      std::vector<long long> boff(bl.size() + 1, 0);
      for (size_t k = 0; k < bl.size(); ++k) boff[k + 1] = boff[k] + blk_svec_len(bl[k]);
      std::vector<int> rows_of(bl.size(), 0), nz_of(bl.size(), 0);
      for (int j = 0; j < m && cand; ++j) {
        if (At_cp[j] == At_cp[j + 1]) continue;
        const long long r0 = At_ri[At_cp[j]];
        const size_t k = (size_t)(std::upper_bound(boff.begin(), boff.end(), r0) - boff.begin()) - 1;
        for (int p = At_cp[j]; p < At_cp[j + 1] && cand; ++p) cand = At_ri[p] >= boff[k] && At_ri[p] < boff[k + 1];
        cand = cand && ++rows_of[k] <= kClosedMaxRows && (nz_of[k] += At_cp[j + 1] - At_cp[j]) <= kFuseRowsMax;
      }
    }
    s->plan.tiny_sign = cand && any_tiny;
    s->closed_candidate = cand;
  }
  rc = s->plan.build(s->blk_local.data(), (int)s->blk_local.size());
  s->plan.overlap = true;
  if (!rc && !s->blk_local.empty() && s->opt_hint != 0) {
    // Schedule warm start of the ONE-WAVEFRONT sign kernels (lift steps each block needed in the previous projection): C2 12.0 ->
    // 11.4 steps (+3 % iterations / s), C4 11.0 -> 10.5.  On the batched-GEMM path for the mid-size blocks only (groups padded to <= 512,
    // psd_large.h; psd_hint = 2: every group): one n = 2 000 block loses a step (C3: 17 -> 18 -- the recorded count includes the overshoot
    // of the previous run's bursts, and a failed first probe costs 4 steps there).
    // The hint ages by one step every 16th projection (PsdPlan::project; inside the task loop of the batched launches).
    if ((rc = s->hint_d.alloc(s->blk_local.size()))) return rc;
    CUADMM_HIP_TRY(hipMemset(s->hint_d.p, 0, sizeof(int) * s->blk_local.size()));
    s->plan.d_hint = s->hint_d.p;
    s->plan.sign.d_hint = s->hint_d.p;
    s->plan.sign.hint_max_n = s->opt_hint == 2 ? (1 << 30) : (s->opt_hint == 3 ? 0 : 512);      // 3: the one-wavefront kernels only (rounds 2 - 4)
  }
  // step counts per block: for cuadmm_get_psd_steps and for the longest-block-first reordering of the fused launches
  if (!rc && (s->psd_steps || s->plan.fusable()) && !s->blk_local.empty()) {
    if ((rc = s->steps_d.alloc(s->blk_local.size()))) return rc;
    CUADMM_HIP_TRY(hipMemset(s->steps_d.p, 0, sizeof(int) * s->blk_local.size()));
    s->plan.d_steps = s->steps_d.p;
    s->plan.sign.d_steps = s->steps_d.p;
  }
  s->plan.sign.allow_graph = true;   // same buffers every iteration: replay the sign-path launch sequence from a hipGraph
  if (rc) return rc;

  return rc;
}

// stage: A^T and A on the device with the permutation folded in; which blocks fuse, which constraint rows are local to one
static int init_matrices(Solver* s, const InitIn& in, InitCtx& c) {
  CUADMM_INIT_STAGE_PROLOGUE
  // --- device matrices with the permutation folded in
  {
    std::vector<int> lrp((size_t)L + 1, 0), lci;
    std::vector<double> lv;
    const int base = rp[s->sv_begin];
    const int cnt = rp[s->sv_end] - base;
    lci.resize(cnt); lv.resize(cnt);
    for (long long i = 0; i <= L; ++i) lrp[i] = rp[s->sv_begin + i] - base;
    for (int q = 0; q < cnt; ++q) { lci[q] = s->perm_inv[rci[base + q]]; lv[q] = rv[base + q]; }
    if ((rc = s->At_long.build(L, lrp.data()))) return rc;
    if ((rc = s->At_rp.from(lrp)) || (rc = s->At_ci.from(lci)) || (rc = s->At_v.from(lv))) return rc;
    // A rows in permuted order, local columns
    std::vector<int> arp((size_t)m + 1, 0), aci;
    std::vector<double> av;
    aci.reserve(cnt); av.reserve(cnt);
    for (int pidx = 0; pidx < m; ++pidx) {
      const int j = s->perm[pidx];
      for (int p = At_cp[j]; p < At_cp[j + 1]; ++p) {
        const long long r = At_ri[p];
        if (r >= s->sv_begin && r < s->sv_end) { aci.push_back((int)(r - s->sv_begin)); av.push_back(vals[p]); }
      }
      arp[pidx + 1] = (int)aci.size();
    }
    s->A_avg_nnz = m > 0 ? (double)aci.size() / m : 1.0;
    if ((rc = s->A_long.build(m, arp.data()))) return rc;
    // fused iteration: needs the one-wavefront-per-block sign kernels and no long rows of A^T (they are summed by their own kernel)
    s->fuse = s->plan.fusable() && s->At_long.nlong == 0 && !s->sw.debug_eig && s->sw.fuse != 0;
    s->lrows.active = false; s->lrows.nlocal = s->lrows.nrest = 0;
    if (s->fuse && s->A_long.nlong == 0 && m > 0 && s->sw.fuse_rows != 0) {
      // constraint rows local to one fused block -> that block's kernel (psd_fuse.h)
      std::vector<int> slot_of;
      s->plan.fused_slots(slot_of);                                    // block -> partial-sum slot, -1: not fused
      const int nslots = s->plan.fused_blocks();
      std::vector<long long> boff((size_t)s->blk_local.size() + 1, 0);
      for (size_t k = 0; k < s->blk_local.size(); ++k) boff[k + 1] = boff[k] + blk_svec_len(s->blk_local[k]);
      std::vector<int> row_slot((size_t)m, -1);
      int nlocal = 0;
      for (int r = 0; r < m; ++r) {
        if (arp[r + 1] == arp[r]) continue;
        const long long c0 = aci[arp[r]];
        const int k = (int)(std::upper_bound(boff.begin(), boff.end(), c0) - boff.begin()) - 1;
        if (slot_of[k] < 0) continue;
        bool in = true;
        for (int p = arp[r]; p < arp[r + 1] && in; ++p) in = aci[p] >= boff[k] && aci[p] < boff[k + 1];
        if (!in) continue;
        row_slot[r] = slot_of[k];
      }
      {   // a block keeps its local rows only if one lane per row and one per nonzero suffice (psd_fuse.h)
        std::vector<long long> nz_of((size_t)nslots, 0), rows_of((size_t)nslots, 0);
        for (int r = 0; r < m; ++r) if (row_slot[r] >= 0) { nz_of[row_slot[r]] += arp[r + 1] - arp[r]; rows_of[row_slot[r]]++; }
        for (int r = 0; r < m; ++r) {
          if (row_slot[r] < 0) continue;
          if (nz_of[row_slot[r]] > kFuseRowsMax || rows_of[row_slot[r]] > kFuseRowsMax) { row_slot[r] = -1; continue; }
          ++nlocal;
        }
      }
      if (nlocal > 0) {
        // rows grouped by BLOCK (plan order of the blocks, rows ascending inside a block); lc[block] = {first row, rows |
        // longest row << 16, first nonzero, nonzeros}
        const size_t nb = s->blk_local.size();
        std::vector<int> blk_of_slot((size_t)nslots, -1);
        for (size_t k = 0; k < nb; ++k) if (slot_of[k] >= 0) blk_of_slot[slot_of[k]] = (int)k;
        std::vector<int> bcnt(nb + 1, 0);
        for (int r = 0; r < m; ++r) if (row_slot[r] >= 0) bcnt[(size_t)blk_of_slot[row_slot[r]] + 1]++;
        for (size_t k = 0; k < nb; ++k) bcnt[k + 1] += bcnt[k];
        std::vector<int> lrow((size_t)nlocal), lnz((size_t)nlocal + 1, 0), le, fill(bcnt.begin(), bcnt.end() - 1);
        std::vector<double> lval;
        for (int r = 0; r < m; ++r) if (row_slot[r] >= 0) lrow[fill[blk_of_slot[row_slot[r]]]++] = r;
        for (int q = 0; q < nlocal; ++q) {
          const int r = lrow[q];
          const long long base = boff[blk_of_slot[row_slot[r]]];
          for (int p = arp[r]; p < arp[r + 1]; ++p) { le.push_back((int)(aci[p] - base)); lval.push_back(av[p]); }
          lnz[q + 1] = (int)le.size();
        }
        std::vector<LcDesc> lcd(nb, LcDesc{0, 0, 0, 0});
        for (size_t k = 0; k < nb; ++k) {
          const int k0 = bcnt[k], k1 = bcnt[k + 1];
          int longest = 0;
          for (int q = k0; q < k1; ++q) longest = std::max(longest, lnz[q + 1] - lnz[q]);
          lcd[k] = LcDesc{k0, (k1 - k0) | (longest << 16), lnz[k0], lnz[k1] - lnz[k0]};
        }
        std::vector<int> rrp{0}, rci, rmap;
        std::vector<double> rv2;
        for (int r = 0; r < m; ++r) {
          if (row_slot[r] >= 0) continue;
          for (int p = arp[r]; p < arp[r + 1]; ++p) { rci.push_back(aci[p]); rv2.push_back(av[p]); }
          rrp.push_back((int)rci.size());
          rmap.push_back(r);
        }
        auto& lr = s->lrows;
        lr.nlocal = nlocal; lr.nrest = (int)rmap.size();
        lr.rest_avg = rmap.empty() ? 1.0 : (double)rci.size() / (double)rmap.size();
        lr.h_desc = lcd; lr.h_row = lrow; lr.h_nzptr = lnz; lr.h_e = le; lr.h_v = lval;
        if ((rc = lr.desc.from(lcd)) || (rc = lr.row.from(lrow)) || (rc = lr.nzptr.from(lnz)) || (rc = lr.e.from(le)) || (rc = lr.v.from(lval))) return rc;
        if (lr.nrest > 0) {
          if ((rc = lr.rest_rp.from(rrp)) || (rc = lr.rest_map.from(rmap)) || (rc = lr.rest_ci.alloc(std::max<size_t>(rci.size(), 1))) ||
              (rc = lr.rest_v.alloc(std::max<size_t>(rv2.size(), 1))) || (rc = lr.rest_ci.upload(rci.data(), rci.size())) ||
              (rc = lr.rest_v.upload(rv2.data(), rv2.size())))
            return rc;
        }
        lr.active = true;
      }
    }
    if (s->A_long.nlong > 0) {   // the average that picks the threads-per-row of the main kernel should not count the capped tails
      long long capped = 0;
      for (int i = 0; i < m; ++i) capped += std::min(arp[i + 1] - arp[i], s->A_long.cap);
      s->A_avg_nnz = (double)capped / m;
    }
    if ((rc = s->A_rp.from(arp)) || (rc = s->A_ci.alloc(std::max<size_t>(aci.size(), 1))) || (rc = s->A_v.alloc(std::max<size_t>(av.size(), 1)))) return rc;
    if ((rc = s->A_ci.upload(aci.data(), aci.size())) || (rc = s->A_v.upload(av.data(), av.size()))) return rc;
    s->A_v.n = av.size();
  }

  return rc;
}

// stage: scaling (solver.cu:169-191), b / C / X / S / y in the solver's units, work vectors
static int init_vectors(Solver* s, const InitIn& in, InitCtx& c) {
  CUADMM_INIT_STAGE_PROLOGUE
  // --- scaling (solver.cu:169-191)
  double nb = 0, nc = 0;
  for (int i = 0; i < b_nnz; ++i) nb += b_vals[i] * b_vals[i];
  for (int i = 0; i < C_nnz; ++i) nc += C_vals[i] * C_vals[i];
  if (s->local_mode) { nb = s->ov_nb; nc = s->ov_nc; }      // norms over ALL constraints / the whole C
  s->norm_borg = 1 + std::sqrt(nb);
  s->norm_Corg = 1 + std::sqrt(nc);
  std::vector<double> bfull(m, 0.0);
  std::vector<char> seen_b((size_t)m, 0);
  double nb2 = 0;
  for (int i = 0; i < b_nnz; ++i) {
    if (b_idx[i] < 0 || b_idx[i] >= m) { set_error("init: b index %d out of range", b_idx[i]); return CUADMM_ERR_INVALID; }
    // a sparse vector with a repeated index has no defined value (the norms would count both entries, the vector one)
    if (seen_b[b_idx[i]]) { set_error("init: b index %d appears twice", b_idx[i]); return CUADMM_ERR_INVALID; }
    seen_b[b_idx[i]] = 1;
    double v = b_vals[i] / s->normA[b_idx[i]];             // sparse_dense.cu:11-20
    bfull[b_idx[i]] = v;
    nb2 += v * v;
  }
  if (s->local_mode) nb2 = s->ov_nb2;
  s->bscale = 1 + std::sqrt(nb2);
  s->Cscale = 1 + std::sqrt(nc);
  s->objscale = s->bscale * s->Cscale;
  const double ibs = 1 / s->bscale, ics = 1 / s->Cscale;   // *_div_scalar multiply by 1/s (dense_scalar.cu:77-81)
  if (s->y_registered) {   // a second init on the same handle: the registration must not outlive the storage it names
    hipError_t e = hipHostUnregister(s->y_p.data()); (void)e;
    s->y_registered = false;
  }
  s->normA_p.resize(m); s->b_p.resize(m); s->y_p.assign(m, 0.0); s->Rp_p.assign(m, 0.0);
  s->y_best_p.assign(m, 0.0);
  if (m > 0 && hipHostRegister(s->y_p.data(), sizeof(double) * (size_t)m, hipHostRegisterDefault) == hipSuccess) s->y_registered = true;
  else { hipError_t e = hipGetLastError(); (void)e; }   // not fatal: upload_y then stages through the pinned buffer
  for (int pidx = 0; pidx < m; ++pidx) {
    const int j = s->perm[pidx];
    s->normA_p[pidx] = s->normA[j];
    s->b_p[pidx] = bfull[j] * ibs;
    if (y0) s->y_p[pidx] = (y0[j] * s->normA[j]) * ics;     // solver.cu:182,191
  }
  {
    std::vector<double> Cl((size_t)L, 0.0);
    std::vector<char> seen_C((size_t)L, 0);
    for (int i = 0; i < C_nnz; ++i) {
      if (C_idx[i] < 0 || C_idx[i] >= vec_len) { set_error("init: C index %d out of range", C_idx[i]); return CUADMM_ERR_INVALID; }
      if (C_idx[i] >= s->sv_begin && C_idx[i] < s->sv_end) {
        if (seen_C[C_idx[i] - s->sv_begin]) { set_error("init: C index %d appears twice", C_idx[i]); return CUADMM_ERR_INVALID; }
        seen_C[C_idx[i] - s->sv_begin] = 1;
        Cl[C_idx[i] - s->sv_begin] = C_vals[i] * ics;
      }
    }
    // kSvecPad: the closed-block kernels read whole batches of the flat svec walk past a block's last element without a
    // bounds clamp (psd_sign_closed.h); the values are never used
    if ((rc = s->C.from(Cl, kSvecPad))) return rc;
    std::vector<double> tmp((size_t)L, 0.0);
    if (X0) for (long long i = 0; i < L; ++i) tmp[i] = X0[s->sv_begin + i] * ibs;
    if ((rc = s->X.from(tmp, kSvecPad))) return rc;
    if (S0) for (long long i = 0; i < L; ++i) tmp[i] = S0[s->sv_begin + i] * ics;
    else std::fill(tmp.begin(), tmp.end(), 0.0);
    if ((rc = s->S.from(tmp, kSvecPad))) return rc;
  }
  if ((rc = s->Rd1.alloc(L)) || (rc = s->Xb.alloc(L)) || (rc = s->Xproj.alloc(L)) || (rc = s->y_d.alloc(std::max(m, 1))) ||
      (rc = s->out_d.alloc(2 * (size_t)m + 2)) || (rc = s->partials.alloc(2 * (size_t)post_grid(L) + 2 + 2 * (size_t)s->plan.fused_blocks())) ||
      (rc = s->h_out.alloc(2 * (size_t)m + 2)) || (rc = s->h_y.alloc(std::max(m, 1))) || (rc = s->scal_d.alloc(8 + 128)) ||
      (rc = s->h_scal.alloc(8)))
    return rc;
  CUADMM_HIP_TRY(hipMemset(s->out_d.p, 0, sizeof(double) * (2 * (size_t)m + 2)));
  std::memset(s->h_out.p, 0, sizeof(double) * (2 * (size_t)m + 2));
  s->out_w = s->out_d.p;
  if (s->fuse && (rc = s->plan.build_rest_index())) return rc;
  if (s->fuse && s->verbose)
    printf(" fused iteration: %d blocks form Xb and apply the S / X updates inside their projection kernel (%lld of %lld svec entries outside); "
           "%d of %d constraint rows are local to one of them\n", s->plan.fused_blocks(), s->plan.n_rest, L, s->lrows.nlocal, m);
  return rc;
}

// stage: where the y-solve runs: device-side sweeps around the GPU tail, one thread per tree of a block-diagonal factor, or the host
static int init_solve_plan(Solver* s, const InitIn& in, InitCtx& c) {
  CUADMM_INIT_STAGE_PROLOGUE
  s->dev_scalars = s->local_mode && s->comm_world > 1 && !s->sw.host_scalars;
  // device-side y-solve: whole factor on the host side of the split (no GPU tail) and a forest of many small trees
  s->dev_solve = false;
  if (s->tail.k > 0 && !s->sw.host_solve) {
    // split factor: leading sweeps on the device too when the leading elimination forest is shallow enough (cost model:
    // the deepest tree decides the kernel; host: 1.2 ns per leading nonzero for both sweeps + the PCIe hops of the tail)
    const int64_t* Lp; const int* Li; const double* Lx; const double* D;
    if ((rc = cuadmm_aat_factor_arrays(s->fac, &Lp, &Li, &Lx, &D))) return rc;
    s->lead.stream_only = s->sw.lead_stream != 0;
    s->lead.pinv_tol = s->sw.pinv_tol;
    s->lead.debug = s->sw.lead_debug != 0;
    s->lead.force_hybrid = s->sw.l21_device == 2;
    s->lead.tops_refine = s->sw.lead_tops_refine != 0;
    s->lead.small_kb = std::max(0, std::min(s->sw.lead_small_kb, 36));
    s->lead.tops_level = s->sw.l21_device == 2 ? 0 : (s->sw.lead_tops >= 0 ? s->sw.lead_tops : (cuadmm_aat_tail_tops(s->fac) > 0 ? cuadmm_aat_tail_tops(s->fac) : -1));
    if ((rc = s->lead.build(m, s->tail.k, Lp, Li, Lx, D, s->sw.l21_device != 0))) return rc;
    const double host_us = 1.2e-3 * (double)Lp[m - s->tail.k] + 150.0;
    if (s->lead.ready && s->lead.tops && !(s->lead.est_us < 0.7 * host_us)) {      // the cut forest loses to the host: the plain paths decide
      s->lead.tops_level = 0;
      if ((rc = s->lead.build(m, s->tail.k, Lp, Li, Lx, D, s->sw.l21_device != 0))) return rc;
    }
    if (s->lead.ready && s->lead.est_us < 0.7 * host_us) {
      s->dev_solve = true;
      if (s->local_mode && s->comm_world > 1) s->dev_scalars = true;
      if (s->verbose && s->lead.tops)
        printf(" y-solve on the device: sweeps over %d trees (depth <= %d), %d dense tree tops (%d columns, %.0f MB of inverses), the GPU tail\n", s->lead.ntrees,
               s->lead.max_levels, s->lead.tops_blocks, s->lead.nT, (double)s->lead.tops_bytes / 1e6);
      else if (s->verbose) printf(" y-solve on the device: leading sweeps over %d trees (depth <= %d) around the GPU tail\n", s->lead.ntrees, s->lead.max_levels);
    } else if (s->lead.hybrid || (s->lead.ready && s->sw.l21_device && s->lead.demote_to_hybrid())) {
      if (s->verbose) printf(" y-solve: L11 sweeps on the host (forest depth %d); L21 (%lld nonzeros) and the tail on the device\n", s->lead.max_levels, s->lead.nnz21);
    } else {
      s->lead.release();
    }
    // the planner sized the tail for the solve with dense tree tops; if that solve could not be built (an inverse failed its check, the
    // rest of the forest is still too deep) what runs instead is correct but slower than round 4's plan -- option lead_tops = 0 restores it
    if (cuadmm_aat_tail_tops(s->fac) > 0 && s->sw.lead_tops < 0 && !(s->dev_solve && s->lead.tops) && s->verbose)
      printf(" y-solve: the tail of %d columns was planned for dense tree tops, which could not be built (%s); option lead_tops = 0 plans without them\n",
             s->tail.k, cuadmm_last_error());
  }
  if (s->tail.k == 0 && m > 0 && !s->sw.host_solve) {
    int ntrees = 0, maxc = 0;
    const int *tp = nullptr, *tc = nullptr;
    if (cuadmm_aat_forest(s->fac, &ntrees, &maxc, &tp, &tc) == CUADMM_OK && ntrees >= 256 && maxc <= 64) {
      const int64_t* Lp; const int* Li; const double* Lx; const double* D;
      if ((rc = cuadmm_aat_factor_arrays(s->fac, &Lp, &Li, &Lx, &D))) return rc;
      const size_t lnz = (size_t)Lp[m];
      std::vector<long long> lp64(Lp, Lp + m + 1);
      if ((rc = s->f_tree_ptr.from(std::vector<int>(tp, tp + ntrees + 1))) || (rc = s->f_tree_cols.from(std::vector<int>(tc, tc + m))) ||
          (rc = s->f_Lp.from(lp64)) || (rc = s->f_Li.from(std::vector<int>(Li, Li + lnz))) || (rc = s->f_Lx.from(std::vector<double>(Lx, Lx + lnz))) ||
          (rc = s->f_D.from(std::vector<double>(D, D + m))))
        return rc;
      s->forest_trees = ntrees;
      s->dev_solve = true;
      if (s->local_mode && s->comm_world > 1) s->dev_scalars = true;
      if (s->verbose) printf(" y-solve on the device: %d independent trees of the elimination forest (<= %d columns each)\n", ntrees, maxc);
    }
  }
  return rc;
}

// stage: closed blocks: per-block records for psd_sign_closed.h
static int init_closed(Solver* s, const InitIn& in, InitCtx& c) {
  CUADMM_INIT_STAGE_PROLOGUE
  s->closed.active = false;
  // Default: only when no block with rows is smaller than 17 -- the solve adds two dependent memory round trips to a block's
  // prologue: < 4 % of the lifetime of an n >= 17 block (C2: 0.307 -> 0.300 ms per iteration, the solve and statistics kernels
  // gone), but 15 % of an n <= 16 block's (C4 with its 67 000 tiny blocks: 1.89 -> 1.91).  CUADMM_FUSE_SOLVE=1 / 0 forces it.
  // (round 2 kept blocks with rows and n < 17 out: the solve added two dependent round trips to their prologue; with the block
  // records of psd_sign_closed.h it rides in the one trip the prologue makes anyway)
  const bool want_closed = s->sw.fuse_solve != 0;
  if (want_closed && s->fuse && s->lrows.active && s->lrows.nrest == 0 && s->dev_solve && !s->lead.ready && s->forest_trees > 0 && s->lrows.nlocal == m) {
    const auto& hd = s->lrows.h_desc;
    const auto& hrow = s->lrows.h_row;
    const size_t nb = hd.size();
    bool ok = true;
    for (const auto& d : hd) ok = ok && (d.y & 0xffff) <= kClosedMaxRows;
    if (ok) {
      const int64_t* Lp; const int* Li; const double* Lx; const double* D;
      if ((rc = cuadmm_aat_factor_arrays(s->fac, &Lp, &Li, &Lx, &D))) return rc;
      std::vector<int> pos((size_t)m, -1), blk_of_row((size_t)m, -1);
      for (size_t k = 0; k < nb; ++k) {
        const int nk = hd[k].y & 0xffff;
        for (int q = 0; q < nk; ++q) { pos[hrow[hd[k].x + q]] = q; blk_of_row[hrow[hd[k].x + q]] = (int)k; }
      }
      // one ClosedRec per fused slot (psd_fuse.h): rows, their b / normA / D, the dense factor, the block's nonzeros of A in
      // local-row order with the round in which each is applied to its svec slot
      std::vector<int> slot_of;
      s->plan.fused_slots(slot_of);
      std::vector<ClosedRec> recs((size_t)std::max(s->plan.fused_blocks(), 1));
      std::memset(recs.data(), 0, sizeof(ClosedRec) * recs.size());
      for (auto& R : recs) for (int a = 0; a < kClosedMaxRows; ++a) R.D[a] = 1.0;     // rows >= nk: the kernel's sweeps run unmasked
      for (size_t k = 0; k < nb && ok; ++k) {
        if (slot_of[k] < 0) continue;
        ClosedRec& R = recs[(size_t)slot_of[k]];
        const int nk = hd[k].y & 0xffff;
        R.nk = nk; R.nnz = hd[k].w;
        if (R.nnz > kFuseRowsMax || nk > kClosedMaxRows) { ok = false; break; }
        for (int a = 0; a < nk && ok; ++a) {
          const int j = hrow[hd[k].x + a];
          R.rows[a] = j;
          R.D[a] = D[j]; R.b[a] = s->b_p[j]; R.normA[a] = s->normA_p[j];
          for (int64_t p = Lp[j]; p < Lp[j + 1]; ++p) {
            const int i = Li[p];
            if (blk_of_row[i] != (int)k || pos[i] <= a) { ok = false; break; }     // fill outside the block: not closed after all
            R.L[pos[i] * kClosedMaxRows + a] = Lx[p];
          }
        }
      }
      if (ok) {
        // nonzeros: s->lrows kept e / v / nzptr on the device only; rebuild them here from the host copies made above
        const auto& le = s->lrows.h_e; const auto& lval = s->lrows.h_v; const auto& lnz = s->lrows.h_nzptr;
        for (size_t k = 0; k < nb; ++k) {
          if (slot_of[k] < 0) continue;
          ClosedRec& R = recs[(size_t)slot_of[k]];
          const int k0 = hd[k].x, nk = R.nk, z0 = hd[k].z;
          int maxmult = 0;
          for (int a = 0; a <= nk; ++a) R.nzp[a] = (unsigned char)(lnz[k0 + a] - z0);
          for (int a = 0; a < nk; ++a) R.maxlen = std::max(R.maxlen, (int)R.nzp[a + 1] - (int)R.nzp[a]);
          const int n_blk = std::abs(s->blk_local[k]);
          for (int q = 0; q < R.nnz; ++q) {
            R.nzt[q] = psd_closed_tab_entry(n_blk, le[z0 + q]);
            R.v[q] = lval[z0 + q];
            int rowpos = 0;
            while (rowpos + 1 <= nk && (int)R.nzp[rowpos + 1] <= q) ++rowpos;
            int mult = 0;
            for (int q2 = 0; q2 < q; ++q2) mult += le[z0 + q2] == le[z0 + q];
            R.rk[q] = (unsigned char)(rowpos | (mult << 3));
            maxmult = std::max(maxmult, mult + 1);
            if (mult > 31) ok = false;
          }
          R.nrounds = maxmult;
        }
      }
      if (ok) {
        if ((rc = s->closed.rec.from(recs)) || (rc = s->closed.cl_out.alloc(16 * recs.size())) ||
            (rc = s->closed.partials2.alloc(2 * (size_t)s->plan.fused_blocks() + 2)))
          return rc;
        CUADMM_HIP_TRY(hipMemset(s->closed.cl_out.p, 0, sizeof(double) * 16 * recs.size()));
        {   // the records' headers travel with the block descriptors (one dependent load less in every kernel that reads them)
          std::vector<int> aux(nb, 0);
          for (size_t k = 0; k < nb; ++k)
            if (slot_of[k] >= 0) { const ClosedRec& R = recs[(size_t)slot_of[k]]; aux[k] = closed_hdr_pack(R.nk, R.nnz, R.nrounds, R.maxlen); }
          if ((rc = s->plan.set_desc_aux(aux))) return rc;
        }
        s->closed.active = true;
        s->closed.out_dirty = true;
        if (s->verbose) printf(" closed blocks: each block solves for its own multipliers (<= %d rows) inside the projection kernel\n", kClosedMaxRows);
      }
    }
  }
  return rc;
}

// stage: device copies of b / normA for the device-side scalars, mapped result buffers, initial residuals (solver.cu:195-228)
static int init_finish(Solver* s, const InitIn& in, InitCtx& c) {
  CUADMM_INIT_STAGE_PROLOGUE
  if (s->dev_scalars || s->dev_solve) {
    if ((rc = s->b_d.alloc(std::max(m, 1))) || (rc = s->normA_d.alloc(std::max(m, 1)))) return rc;
    if ((rc = s->b_d.upload(s->b_p.data(), (size_t)m)) || (rc = s->normA_d.upload(s->normA_p.data(), (size_t)m))) return rc;
    void* dp = nullptr;
    if (s->sw.mapped_out && hipHostGetDevicePointer(&dp, s->h_scal.p, 0) == hipSuccess && dp) s->h_scal_dev = static_cast<double*>(dp);
    else { hipError_t e = hipGetLastError(); (void)e; }
  }
  if (s->world <= 1 && !s->force_comm && !s->dev_scalars && !s->dev_solve && s->sw.mapped_out) {
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, s->h_out.p, 0) == hipSuccess && dp) { s->out_w = static_cast<double*>(dp); s->out_mapped = true; }
    else { hipError_t e = hipGetLastError(); (void)e; }
  }

  // --- initial residuals (solver.cu:195-228)
  if ((rc = s->upload_y(true))) return rc;
  if ((rc = s->launch_aty(false))) return rc;                               // Rd1 = At*y - C
  if ((rc = launch_post(2, L, s->Xproj.p, s->Rd1.p, s->C.p, s->X.p, s->S.p, 1.0, 0.0, s->partials.p,
                        s->out_w + (size_t)m, s->st)))                      // Rd = Rd1 + S, sums (X untouched)
    return rc;
  if ((rc = s->launch_spmv(true, true))) return rc;
  if ((rc = s->fetch_out(0, 2 * (size_t)m + 2))) return rc;
  {
    double nr = 0, bty = 0;
    for (int i = 0; i < m && !s->dev_solve; ++i) {
      s->Rp_p[i] = -s->h_out.p[i] + s->b_p[i];
      double ro = s->normA_p[i] * s->Rp_p[i] * s->bscale;
      nr += ro * ro;
      bty += s->b_p[i] * s->y_p[i];
    }
    {
      double v[4] = {nr, bty, s->h_out.p[(size_t)m], s->h_out.p[(size_t)m + 1]};
      if (s->dev_scalars || s->dev_solve) { for (int q = 0; q < 4; ++q) v[q] = s->h_scal.p[q]; }   // formed (and summed over ranks) on the device
      else if ((rc = s->allreduce_scalars(v, 4))) return rc;
      nr = v[0]; bty = v[1]; s->h_out.p[(size_t)m] = v[2]; s->h_out.p[(size_t)m + 1] = v[3];
    }
    s->errRp = std::sqrt(nr) / s->norm_borg;
    s->errRd = std::sqrt(s->h_out.p[(size_t)m]) * s->Cscale / s->norm_Corg;
    s->maxfeas = std::max(s->errRp, s->errRd);
    s->pobj = s->h_out.p[(size_t)m + 1] * s->objscale;
    s->dobj = bty * s->objscale;
    s->relgap = std::fabs(s->pobj - s->dobj) / (1 + std::fabs(s->pobj) + std::fabs(s->dobj));
  }
  return rc;
}

#undef CUADMM_INIT_STAGE_PROLOGUE
}  // namespace

extern "C" {

const char* cuadmm_last_error(void) { return get_error(); }
const char* cuadmm_version(void) { return "cuadmm_amd 0.1 (gfx950)"; }
int cuadmm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int cuadmm_create(cuadmm_solver** out) {
  if (!out) { set_error("create: null"); return CUADMM_ERR_INVALID; }
  *out = new cuadmm_solver();
  return CUADMM_OK;
}
void cuadmm_destroy(cuadmm_solver* s) {
  if (s && s->group) { duo_group_destroy(s->group); s->group = nullptr; }
  delete s;
}

int cuadmm_set_option(cuadmm_solver* s, const char* key, double value) {
  if (!s || !key) { set_error("set_option: null"); return CUADMM_ERR_INVALID; }
  std::string k(key);
  s->option_log.emplace_back(k, value);
  if (k == "device") s->device = (int)value;
  else if (k == "verbose") s->verbose = (int)value;
  else if (k == "rank") s->rank = (int)value;
  else if (k == "world") s->world = (int)value;
  else if (k == "profile") s->profile = (int)value;
  else if (k == "force_comm") s->force_comm = (int)value;   // call the collective hook even when world == 1 (testing)
  else if (k == "eig_rank") s->eig_rank = (int)value;
  else if (k == "eig_rank_begin_iter") s->eig_rank_begin_iter = (int)value;
  else if (k == "eig_rank_maxfeas") s->eig_rank_maxfeas = value;
  else if (k == "psd_steps") s->psd_steps = (int)value;       // record the sign kernels' step count per block (cuadmm_get_psd_steps)
  else if (k == "batch") s->bt.max_iters = std::max(0, std::min(256, (int)value));   // iterations per launch, closed blocks (0 / 1: off)
  else if (k == "fuse") s->sw.fuse = (int)value;
  else if (k == "fuse_rows") s->sw.fuse_rows = (int)value;
  else if (k == "fuse_solve") s->sw.fuse_solve = (int)value;
  else if (k == "host_solve") s->sw.host_solve = (int)value;
  else if (k == "host_scalars") s->sw.host_scalars = (int)value;
  else if (k == "tail_k") s->sw.tail_k = (int)value;
  else if (k == "local_constraints") s->sw.local_constraints = (int)value;
  else if (k == "mapped_out") s->sw.mapped_out = (int)value;
  else if (k == "lpt") s->sw.lpt = (int)value;
  else if (k == "aty_post2") s->sw.aty_post2 = (int)value;
  else if (k == "lead_stream") s->sw.lead_stream = (int)value;
  else if (k == "lead_debug") s->sw.lead_debug = (int)value;
  else if (k == "lead_tops") s->sw.lead_tops = (int)value;
  else if (k == "lead_tops_refine") s->sw.lead_tops_refine = (int)value;
  else if (k == "lead_small_kb") s->sw.lead_small_kb = (int)value;
  else if (k == "tail_pinv_tol") s->sw.tail_pinv_tol = value;
  else if (k == "pinv_tol") s->sw.pinv_tol = value;
  else if (k == "l21_device") s->sw.l21_device = (int)value;
  else if (k == "tail_one_pass") s->sw.tail_one_pass = (int)value;
  else if (k == "tail_shard") s->sw.tail_shard = (int)value;
  else if (k == "tail_prefetch") s->sw.tail_prefetch = (int)value;
  else if (k == "tail_depth") { if (value < 0 || value > 3) { set_error("set_option: tail_depth must be 0 .. 3"); return CUADMM_ERR_INVALID; } s->sw.tail_depth = (int)value; }
  else if (k == "tail_order") s->sw.tail_order = value == 2 ? 2 : (value != 0 ? 1 : 0);
  else if (k == "tail_rb") s->sw.tail_rb = (int)value;
  else if (k == "tail_zreg") s->sw.tail_zreg = (int)value;
  else if (k == "tail_group_pf") s->sw.tail_group_pf = (int)value;
  else if (k == "tail_fat") s->sw.tail_fat = (int)value;
  else if (k == "tail_dd") s->sw.tail_dd = (int)value;
  else if (k == "tail_refine") s->sw.tail_refine = (int)value;
  else if (k == "tail_pivot") s->sw.tail_pivot = (int)value;
  else if (k == "tail_max_k") s->sw.tail_max_k = (int)value;
  else if (k == "solve_next") s->sw.solve_next = (int)value;
  else if (k == "debug_eig") s->sw.debug_eig = (int)value;
  else if (s->plan.opt.set(k, value)) {}                      // "psd_*": the projection planner's switches (psd_options.h)
  else if (k == "batch_mixed") s->bt.allow_mixed = value != 0;
  else if (k == "lazy_unscale") s->lazy_unscale = (int)value;
  else if (k == "psd_hint") s->opt_hint = (int)value;
  else if (k == "duo_cpu_eig_on_gpu") s->duo_cpu_eig_on_gpu = (int)value;
  else if (k == "duo_share_device") s->duo_share_device = (int)value;   // duo_init(device_num_requested = N) from one process: all N engines on this solver's device
  else if (k == "tiny_sign") s->opt_tiny_sign = (int)value;                               // n <= 8 on the sign kernel (before init)                                     // schedule warm start (before init)                            // 0: unscale X, y, S at the end of every solve
  else if (k == "duo_exchange") s->duo_exchange = (int)value;           // in-process group: -1 choose, 0 host-staged, 1 device-side exchange
  else if (k == "duo_inject_fail") { s->duo_inject = (long long)value; if (s->group) duo_group_inject(s->group, s->duo_inject); }   // test hook
  else if (k == "graph") {}
  else { s->option_log.pop_back(); set_error("set_option: unknown key '%s'", key); return CUADMM_ERR_INVALID; }
  // a group handle: every rank follows -- AFTER the key has been validated on the leader; a child that refuses leaves the option
  // out of the log (it is not replayed on later children) and the error with the caller
  if (s->group && !s->in_group_call && k != "device" && k != "rank" && k != "world" && k != "verbose" && k != "duo_inject_fail")
    for (int r = 1; r < duo_group_world(s->group); ++r) {
      int rc = cuadmm_set_option(duo_group_rank(s->group, r), key, value);
      if (rc) { s->option_log.pop_back(); return rc; }
    }
  return CUADMM_OK;
}

int cuadmm_set_allreduce(cuadmm_solver* s, cuadmm_allreduce_fn fn, void* user) {
  if (!s) { set_error("set_allreduce: null"); return CUADMM_ERR_INVALID; }
  s->allreduce = fn; s->allreduce_user = user;
  return CUADMM_OK;
}

int cuadmm_rccl_unique_id(char out128[128]) {
  if (!g_rccl.load()) { set_error("RCCL not loadable"); return CUADMM_ERR_COMM; }
  NcclUid id;
  if (g_rccl.GetUniqueId(&id)) { set_error("ncclGetUniqueId failed"); return CUADMM_ERR_COMM; }
  std::memcpy(out128, id.internal, 128);
  return CUADMM_OK;
}

int cuadmm_use_rccl(cuadmm_solver* s, const char unique_id128[128], int rank, int world) {
  if (!s || !unique_id128) { set_error("use_rccl: null"); return CUADMM_ERR_INVALID; }
  if (!g_rccl.load()) { set_error("RCCL not loadable"); return CUADMM_ERR_COMM; }
  int rc = check_device(s->device);
  if (rc) return rc;
  NcclUid id;
  std::memcpy(id.internal, unique_id128, 128);
  if (g_rccl.CommInitRank(&s->rccl_comm, world, id, rank)) { set_error("ncclCommInitRank failed"); return CUADMM_ERR_COMM; }
  s->rank = rank; s->world = world;
  return CUADMM_OK;
}

// ------------------------------------------------------------------------------------------
// SDPSolver::init
// ------------------------------------------------------------------------------------------
int cuadmm_init(cuadmm_solver* s, int eig_stream_num_per_gpu, int cpu_eig_thread_num, int vec_len, int con_num,
                const int* At_cp, const int* At_ri, const double* At_vx, int At_nnz, const int* b_idx,
                const double* b_vals, int b_nnz, const int* C_idx, const double* C_vals, int C_nnz,
                const int* blk, int mat_num, const double* X0, const double* y0, const double* S0, double sig) {
  (void)eig_stream_num_per_gpu; (void)cpu_eig_thread_num;
  if (!s) { set_error("init: null solver"); return CUADMM_ERR_INVALID; }
  if (s->initialised) { set_error("init: solver already initialised (one init per object, as in the reference)"); return CUADMM_ERR_INVALID; }
  if (vec_len < 0 || con_num < 0 || At_nnz < 0 || b_nnz < 0 || C_nnz < 0 || mat_num < 0 || !At_cp || !blk ||
      (At_nnz > 0 && (!At_ri || !At_vx)) || (b_nnz > 0 && (!b_idx || !b_vals)) || (C_nnz > 0 && (!C_idx || !C_vals))) {
    set_error("init: invalid argument");
    return CUADMM_ERR_INVALID;
  }
  if (s->world < 1 || s->rank < 0 || s->rank >= s->world) { set_error("init: bad rank/world %d/%d", s->rank, s->world); return CUADMM_ERR_INVALID; }
  long long Lchk = 0;
  for (int k = 0; k < mat_num; ++k) {
    if (blk[k] == 0) { set_error("init: block %d has size 0", k); return CUADMM_ERR_INVALID; }
    Lchk += blk_svec_len(blk[k]);                       // negative size = unconstrained block of -blk[k] variables
  }
  if (Lchk != vec_len) { set_error("init: vec_len %d does not match blk (sum n(n+1)/2 = %lld)", vec_len, Lchk); return CUADMM_ERR_INVALID; }
  if (At_cp[0] != 0 || At_cp[con_num] != At_nnz) { set_error("init: At column pointers inconsistent with At_nnz"); return CUADMM_ERR_INVALID; }
  for (int p = 0; p < At_nnz; ++p)
    if (At_ri[p] < 0 || At_ri[p] >= vec_len) { set_error("init: At row index %d out of range at %d", At_ri[p], p); return CUADMM_ERR_INVALID; }

  if (s->world > 1) cuadmm_host_pool_hint(s->world);       // the ranks of a job share the node's CPUs (before the pool's first use)
  if (s->world > 1 && !s->local_mode && s->sw.local_constraints) {
    // does every constraint live inside one rank's block range?
    std::vector<int> first;
    partition_blocks(blk, mat_num, s->world, first);
    std::vector<long long> svb((size_t)s->world + 1, 0);
    {
      long long off = 0;
      int r = 0;
      for (int k = 0; k <= mat_num; ++k) {
        while (r <= s->world && first[r] == k) svb[r++] = off;
        if (k < mat_num) off += blk_svec_len(blk[k]);
      }
    }
    std::vector<int> owner(con_num, 0);
    bool all_owned = true;
    for (int j = 0; j < con_num && all_owned; ++j) {
      if (At_cp[j] == At_cp[j + 1]) continue;                      // empty constraint: rank 0
      const int r = (int)(std::upper_bound(svb.begin(), svb.end(), (long long)At_ri[At_cp[j]]) - svb.begin()) - 1;
      for (int p = At_cp[j]; p < At_cp[j + 1]; ++p)
        if (At_ri[p] < svb[r] || At_ri[p] >= svb[r + 1]) { all_owned = false; break; }
      owner[j] = r;
    }
    if (all_owned) {
      const int me = s->rank;
      // global norms (solver.cu:169-191) from the full inputs every rank holds
      double nb = 0, nc = 0, nb2 = 0;
      for (int i = 0; i < b_nnz; ++i) {
        if (b_idx[i] < 0 || b_idx[i] >= con_num) { set_error("init: b index %d out of range", b_idx[i]); return CUADMM_ERR_INVALID; }
        nb += b_vals[i] * b_vals[i];
        double cn = 0;
        for (int p = At_cp[b_idx[i]]; p < At_cp[b_idx[i] + 1]; ++p) cn += At_vx[p] * At_vx[p];
        const double v = b_vals[i] / std::max(1.0, std::sqrt(cn));
        nb2 += v * v;
      }
      for (int i = 0; i < C_nnz; ++i) nc += C_vals[i] * C_vals[i];
      std::vector<int> cons, g2l(con_num, -1);
      for (int j = 0; j < con_num; ++j) if (owner[j] == me) { g2l[j] = (int)cons.size(); cons.push_back(j); }
      const long long lo = svb[me], hi = svb[me + 1];
      std::vector<int> lcp(cons.size() + 1, 0), lri, lbi, lCi;
      std::vector<double> lvx, lbv, lCv, ly0;
      for (size_t q = 0; q < cons.size(); ++q) {
        for (int p = At_cp[cons[q]]; p < At_cp[cons[q] + 1]; ++p) { lri.push_back((int)(At_ri[p] - lo)); lvx.push_back(At_vx[p]); }
        lcp[q + 1] = (int)lri.size();
      }
      for (int i = 0; i < b_nnz; ++i) if (g2l[b_idx[i]] >= 0) { lbi.push_back(g2l[b_idx[i]]); lbv.push_back(b_vals[i]); }
      for (int i = 0; i < C_nnz; ++i) {
        if (C_idx[i] < 0 || C_idx[i] >= vec_len) { set_error("init: C index %d out of range", C_idx[i]); return CUADMM_ERR_INVALID; }
        if (C_idx[i] >= lo && C_idx[i] < hi) { lCi.push_back((int)(C_idx[i] - lo)); lCv.push_back(C_vals[i]); }
      }
      if (y0) { ly0.resize(cons.size()); for (size_t q = 0; q < cons.size(); ++q) ly0[q] = y0[cons[q]]; }
      s->local_mode = true;
      s->comm_world = s->world; s->comm_rank = s->rank;
      s->m_full = con_num; s->cons_local = cons; s->sv_off = lo; s->blk_off = first[me];
      s->L_caller = vec_len; s->nblk_caller = mat_num;
      s->ov_nb = nb; s->ov_nc = nc; s->ov_nb2 = nb2;
      s->world = 1; s->rank = 0;
      if (s->comm_rank != 0) s->verbose = 0;      // one console table per job
      int one = 0;
      return cuadmm_init(s, eig_stream_num_per_gpu, cpu_eig_thread_num, (int)(hi - lo), (int)cons.size(), lcp.data(),
                         lri.empty() ? &one : lri.data(), lvx.empty() ? nullptr : lvx.data(), (int)lri.size(),
                         lbi.empty() ? nullptr : lbi.data(), lbv.empty() ? nullptr : lbv.data(), (int)lbi.size(),
                         lCi.empty() ? nullptr : lCi.data(), lCv.empty() ? nullptr : lCv.data(), (int)lCi.size(),
                         blk + first[me], first[me + 1] - first[me], X0 ? X0 + lo : nullptr, y0 ? ly0.data() : nullptr,
                         S0 ? S0 + lo : nullptr, sig);
    }
  }

  InitIn in{vec_len, con_num, At_nnz, b_nnz, C_nnz, mat_num, At_cp, At_ri, b_idx, C_idx, blk, At_vx, b_vals, C_vals, X0, y0, S0, sig};
  InitCtx ctx;
  int rc;
  if ((rc = init_device(s, in, ctx)) || (rc = init_normalise(s, in, ctx)) || (rc = init_factor(s, in, ctx)) || (rc = init_plan(s, in, ctx)) ||
      (rc = init_matrices(s, in, ctx)) || (rc = init_vectors(s, in, ctx)) || (rc = init_solve_plan(s, in, ctx)) || (rc = init_closed(s, in, ctx)) ||
      (rc = init_finish(s, in, ctx)))
    return rc;
  s->prim_win = 0; s->dual_win = 0; s->ratioconst = 1e0; s->sigmax = 1e3; s->sigmin = 1e-3;
  s->initialised = true;
  return CUADMM_OK;
}

// ------------------------------------------------------------------------------------------
// SDPSolver::solve
// ------------------------------------------------------------------------------------------
int cuadmm_solve(cuadmm_solver* s, int max_iter, double stop_tol, int sig_update_threshold, int sig_update_stage_1,
                 int sig_update_stage_2, int switch_admm, double sigscale, int if_first) {
  if (!s || !s->initialised) { set_error("solve: solver not initialised"); return CUADMM_ERR_INVALID; }
  if (sig_update_stage_1 <= 0 || sig_update_stage_2 <= 0) { set_error("solve: sig_update stages must be positive"); return CUADMM_ERR_INVALID; }
  if (s->group && !s->in_group_call) {    // the leader of an in-process group (duo_group.hip): every rank solves, on its own host thread
    DeviceGuard keep_device;
    return duo_group_run(s->group, [&](cuadmm_solver* q, int) {
      q->in_group_call = true;
      int r2 = cuadmm_solve(q, max_iter, stop_tol, sig_update_threshold, sig_update_stage_1, sig_update_stage_2, switch_admm, sigscale, if_first);
      q->in_group_call = false;
      return r2;
    });
  }
  int rc = check_device(s->device);
  if (rc) return rc;
  s->y_early = false;                       // (a solve that ended in an error may have left it set)
  // the dense tail of the replicated y-solve: split by rows over the ranks of a sharded engine (tail_solve.h)
  if (s->tail.k > 0) {
    const bool shard = s->world > 1 && !s->local_mode && s->sw.tail_shard != 0;
    s->tail.shard_rank = shard ? s->rank : 0;
    s->tail.shard_world = shard ? s->world : 1;
    s->tail.reduce_user = s;
    s->tail.reduce_fn = [](void* user, double* buf, size_t count, hipStream_t) -> int { return static_cast<cuadmm_solver*>(user)->comm_allreduce(buf, count); };
  }
  const int m = s->m;
  const long long L = s->L;
  const bool verbose = s->verbose && s->rank == 0;
  bool breakyes = false;
  std::string final_msg;
  s->info_iter_num = 0;

  if (verbose) {
    printf("\n -------------------------------------------------------------------------------");
    printf("\n                                    cuADMM");
    printf("\n -------------------------------------------------------------------------------");
    printf("\n norm of C = %2.1e, norm of b = %2.1e\n", s->norm_Corg, s->norm_borg);
  }

  if (if_first && s->pending_unscale && (rc = s->materialise())) return rc;   // the reference's X, y, S are unscaled after a solve
  if (!if_first && s->pending_unscale) {
    // the previous solve left X, y, S scaled and [A X | A (S - C)], Rp as its last iteration formed them: nothing to redo
    s->pending_unscale = false;
  } else if (!if_first) {   // solver.cu:385-409: X,y,S currently hold UNSCALED values
    for (int i = 0; i < m; ++i) s->y_p[i] = (s->y_p[i] * s->normA_p[i]) * (1 / s->Cscale);
    if (s->dev_solve && (rc = s->upload_y(true))) return rc;
    if ((rc = launch_scale(s->X.p, L, 1 / s->bscale, s->st))) return rc;
    if ((rc = launch_scale(s->S.p, L, 1 / s->Cscale, s->st))) return rc;
    if ((rc = s->launch_spmv(true, true))) return rc;
    if ((rc = s->fetch_out(0, 2 * (size_t)m + 2))) return rc;
    for (int i = 0; i < m && !s->dev_solve; ++i) s->Rp_p[i] = -s->h_out.p[i] + s->b_p[i];
  }

  if (verbose) {
    std::cout << std::endl << "  it. | p infeas d infeas | primal obj.   dual obj. rel. gap |  time |   sigma | " << std::endl;
    std::cout << " -------------------------------------------------------------------------------" << std::endl;
  }

  if ((rc = s->batch_agree())) return rc;
  const bool lpt_enabled = s->sw.lpt != 0;
  long long& lpt_ev = s->lpt_next;             // counts iterations over all solve calls of this solver
  if (!lpt_enabled) lpt_ev = 0;
  for (int iter = 1; iter <= max_iter + 1; ++iter) {
    // ---- Step 0 (solver.cu:419-467)
    if (std::max(s->maxfeas, s->relgap) < stop_tol) { breakyes = true; final_msg = "Solver ended: converged."; }
    if (iter > max_iter) { breakyes = true; final_msg = "Solver ended: maximum iteration reached"; }
    const double seconds = wall_s() - s->t_init0;
    if (verbose && (breakyes || (iter <= 200 && iter % 50 == 1) || (iter > 200 && iter % 100 == 1))) {
      printf(" %4d | %3.2e %3.2e | %- 5.4e %- 5.4e %3.2e | %5.1f | %2.1e |", iter - 1, s->errRp, s->errRd, s->pobj,
             s->dobj, s->relgap, seconds, s->sig);
      std::cout << std::endl;
    }
    if (breakyes) {
      if (verbose) {
        printf("\n -------------------------------------------------------------------------------\n\n");
        std::cout << final_msg << std::endl;
        printf("\n primal infeasibility = %2.1e \n dual   infeasibility = %2.1e \n relative gap         = %2.1e", s->errRp,
               s->errRd, s->relgap);
        printf("\n primal objective = %- 9.8e \n dual   objective = %- 9.8e", s->pobj, s->dobj);
        printf("\n\n time per iteration = %2.4fs \n total time         = %2.1fs", seconds / iter, seconds);
        printf("\n -------------------------------------------------------------------------------\n\n");
      }
      s->total_time = wall_s() - s->t_init0;
    }

    // the device is ahead of the host schedule by the unconsumed iterations of a batch: back to the accepted state
    if (breakyes && s->bt.len > 0 && (rc = s->batch_rollback(s->bt.pos))) return rc;

    // ---- Step 1 (solver.cu:478-500): y = (AA^T)^-1 (Rp/sig - A(S-C)); with closed blocks the fused projection of this
    // iteration solves it (not on the last pass through the loop, which stops before the projection)
    if (s->y_early) s->y_early = false;       // enqueued at the end of the previous iteration (fetch_out, solve_next)
    else if (s->bt.len == 0 && (rc = s->host_solve(s->fuse && !breakyes))) return rc;

    if (breakyes) {   // solver.cu:567-576
      if (iter > switch_admm && s->have_best) {
        CUADMM_HIP_TRY(hipMemcpyAsync(s->X.p, s->X_best.p, sizeof(double) * (size_t)L, hipMemcpyDeviceToDevice, s->st));
        CUADMM_HIP_TRY(hipMemcpyAsync(s->S.p, s->S_best.p, sizeof(double) * (size_t)L, hipMemcpyDeviceToDevice, s->st));
        CUADMM_HIP_TRY(hipStreamSynchronize(s->st));
        if (s->dev_solve) CUADMM_HIP_TRY(hipMemcpy(s->y_d.p, s->y_best_d.p, sizeof(double) * (size_t)m, hipMemcpyDeviceToDevice));
        else std::copy(s->y_best_p.begin(), s->y_best_p.end(), s->y_p.begin());   // in place: y_p's storage is page-locked
        if (verbose) printf("best max KKT residual after switch  = %2.1e \n", s->best_KKT);
      }
      break;
    }

    // ---- Step 2 (solver.cu:514-656).  The host-side decisions of this iteration (tau, snapshot) depend only on the previous
    // iteration's scalars, so they are taken first: the fused projection needs to know which post step follows it.
    double tau = (iter < switch_admm) ? 1.95 : 1.618;                    // solver.cu:747-754
    if (s->errRd < stop_tol) tau = std::max(1.618, tau / 1.1);

    // does this iteration snapshot the iterate between the S update and the X update?
    bool snapshot = false;
    if (iter == switch_admm) {                                           // solver.cu:681-690
      if (verbose) std::cout << " switching to normal ADMM!" << std::endl;
      sig_update_stage_2 = sig_update_stage_2 / 2;
      if (sig_update_stage_2 < 1) sig_update_stage_2 = 1;
      sigscale = sigscale * 1.23;
      s->sgs_KKT = std::max(s->maxfeas, s->relgap);
      s->best_KKT = s->sgs_KKT;
      snapshot = true;
    } else if (iter > switch_admm && s->have_best && s->best_KKT > std::max(s->maxfeas, s->relgap)) {
      s->best_KKT = std::max(s->maxfeas, s->relgap);                     // solver.cu:732-741
      snapshot = true;
    }

    const int post_mode_after_proj = (iter < switch_admm || snapshot) ? 1 : 0;
    // ---- several iterations in one launch (SignFuse::iters): ADMM phase without best-iterate bookkeeping, every block closed.
    // A batch ends at the next iteration that may change sigma; tau changes only through the errRd < stop_tol rule, checked
    // below on every consumed iteration.
    if (s->bt.len > 0 && tau != s->bt.tau) {
      // this iteration was run with the wrong step length: keep the consumed ones, continue one launch at a time
      if ((rc = s->batch_rollback(s->bt.pos))) return rc;
    }
    if (s->bt.len == 0 && iter > switch_admm && !s->have_best && !snapshot && s->can_batch()) {
      if ((rc = s->batch_alloc())) return rc;
      int K = std::min(std::min(s->bt.max_iters, s->bt.cap), max_iter - iter + 1);
      for (int j = 0; j < K; ++j) {          // the batch may END on a sigma-update iteration, not contain one
        const int i = iter + j;
        if ((i <= sig_update_threshold && i % sig_update_stage_1 == 1) || (i > sig_update_threshold && i % sig_update_stage_2 == 1)) { K = j + 1; break; }
      }
      if (K >= 2) {
        // a batch can only be invalidated by the stopping test or the errRd < stop_tol rule for tau: no checkpoint without a tolerance
        s->bt.have_ck = stop_tol > 0.0;
        if (s->bt.have_ck && (rc = s->batch_copy(true))) return rc;
        s->bt.tau = tau; s->bt.sig = s->sig;
        // longest block first without a stall: the step counts are fetched behind one batch; the next one is launched in the new
        // order (the copy of the sorted descriptors is queued on the stream ahead of the launch; the host sort takes ~0.1 ms)
        const bool lpt_sorted_now = lpt_ev > 0 && s->steps_d.p && s->lpt_steps_ready;
        if (lpt_sorted_now) {
          if ((rc = s->plan.reorder_by_steps_async(s->steps_pin.p, s->st))) return rc;
          s->lpt_steps_ready = false;
          lpt_ev *= 8;
        }
        if ((rc = s->launch_fused_step(0, tau, K))) return rc;
        if (lpt_ev > 0 && s->steps_d.p && !lpt_sorted_now) {
          if (s->lpt_iters + K >= lpt_ev) {
            if (!s->steps_pin.p && (rc = s->steps_pin.alloc(s->steps_d.n))) return rc;
            CUADMM_HIP_TRY(hipMemcpyAsync(s->steps_pin.p, s->steps_d.p, sizeof(int) * s->steps_d.n, hipMemcpyDeviceToHost, s->st));
            s->lpt_steps_ready = true;     // valid once the stream has been synchronised (below)
          }
        }
        CUADMM_HIP_TRY(hipStreamSynchronize(s->st));
        s->prof_collect();
        s->bt.len = K; s->bt.pos = 0;
      }
    }
    const bool from_batch = s->bt.len > 0;
    // the next iteration's y-solve may follow this iteration's last kernel at once (fetch_out, solve_next): the whole solve is on the
    // device and belongs to the engine (not to a closed block's kernel), sigma does not change in this iteration's step 5, nothing of a
    // batch is pending, and the per-class event timers (profile = 1) are off -- they are collected at the wait, before that solve ends
    const bool sig_may_change = (iter <= sig_update_threshold && iter % sig_update_stage_1 == 1) || (iter > sig_update_threshold && iter % sig_update_stage_2 == 1);
    const bool solve_next_ok = s->sw.solve_next != 0 && s->dev_solve && !from_batch && !sig_may_change && !(s->fuse && s->closed.active) && s->profile != 1;
    if (from_batch) {
      // consumed below (Step 5) from bt.h
    } else if ((rc = s->upload_y())) return rc;
    if (from_batch) {
    } else if (s->fuse) {
      if ((rc = s->launch_fused_step(post_mode_after_proj, tau))) return rc;
    } else {
      if ((rc = s->launch_aty(true))) return rc;
      s->plan.rank_active = s->eig_rank > 0 && (iter >= s->eig_rank_begin_iter || s->maxfeas < s->eig_rank_maxfeas);   // duo_solver.cu:844
      if ((rc = s->launch_project())) return rc;
      const char* const debug_eig_dir = s->sw.debug_eig ? getenv("CUADMM_DEBUG_EIG") : nullptr;
      if (debug_eig_dir) {   // developer aid (option debug_eig + CUADMM_DEBUG_EIG=<dir>): dump the projection input when a block hits the QL cap
        int f = s->plan.fail_count(s->st);
        if (f != s->eig_fail_total) {
          fprintf(stderr, "[cuadmm debug] iter %d: QL cap hits %d -> %d\n", iter, s->eig_fail_total, f);
          std::vector<double> h((size_t)L);
          if (staged_d2h(h.data(), s->Xb.p, sizeof(double) * (size_t)L, s->st) == CUADMM_OK) {
            char fn[256];
            snprintf(fn, sizeof fn, "%s/xb_fail_iter%d.bin", debug_eig_dir, iter);
            if (FILE* fp = fopen(fn, "wb")) { fwrite(h.data(), sizeof(double), (size_t)L, fp); fclose(fp); }
          }
          s->eig_fail_total = f;
        }
      }
    }

    if (from_batch) {
    } else if (iter < switch_admm) {
      // sGS half step: S^{k+1}, second solve with it, Rd1 from the new y (solver.cu:693-729)
      if (!s->fuse && (rc = s->launch_post_mode(1, tau))) return rc;     // fused: done with the projection
      if ((rc = s->launch_spmv(false, true, s->fuse))) return rc;
      if ((rc = s->fetch_out((size_t)m + 2, (size_t)m))) return rc;
      if ((rc = s->host_solve())) return rc;
      if ((rc = s->upload_y())) return rc;
      if (s->At_long.nlong == 0 && s->sw.aty_post2) {   // one pass, Rd1 not stored (vec_kernels.hip)
        s->prof_begin(K_POST);
        rc = launch_aty_post2(L, s->At_rp.p, s->At_ci.p, s->At_v.p, s->y_d.p, s->C.p, s->S.p, s->X.p, tau * s->sig, s->partials.p,
                              s->out_w + (size_t)m, s->st);
        s->prof_end(K_POST, 40.0 * (double)L);
        if (rc) return rc;
      } else {
        if ((rc = s->launch_aty(false))) return rc;
        if ((rc = s->launch_post_mode(2, tau))) return rc;
      }
      if ((rc = s->launch_spmv(true, false))) return rc;
      if ((rc = s->fetch_out(0, (size_t)m + 2, solve_next_ok))) return rc;   // [A*X | sums]; A*(S-C) unchanged since the half step
    } else {
      if (snapshot) {
        if (!s->X_best.p && L > 0) { if ((rc = s->X_best.alloc(L)) || (rc = s->S_best.alloc(L))) return rc; }
        if (!s->fuse && (rc = s->launch_post_mode(1, tau))) return rc;
        s->prof_begin(K_COPY);
        CUADMM_HIP_TRY(hipMemcpyAsync(s->X_best.p, s->X.p, sizeof(double) * (size_t)L, hipMemcpyDeviceToDevice, s->st));
        CUADMM_HIP_TRY(hipMemcpyAsync(s->S_best.p, s->S.p, sizeof(double) * (size_t)L, hipMemcpyDeviceToDevice, s->st));
        s->prof_end(K_COPY, 32.0 * (double)L);
        if (s->dev_solve) {
          if (!s->y_best_d.p && (rc = s->y_best_d.alloc(std::max(m, 1)))) return rc;
          CUADMM_HIP_TRY(hipMemcpyAsync(s->y_best_d.p, s->y_d.p, sizeof(double) * (size_t)m, hipMemcpyDeviceToDevice, s->st));
        } else {
          s->y_best_p = s->y_p;
        }
        s->have_best = true;
        if ((rc = s->launch_post_mode(2, tau))) return rc;
      } else if (!s->fuse) {
        if ((rc = s->launch_post_mode(0, tau))) return rc;
      }
      if ((rc = s->launch_spmv(true, true, s->fuse && !snapshot))) return rc;
      if ((rc = s->fetch_out(0, 2 * (size_t)m + 2, solve_next_ok))) return rc;
    }

    // ---- Step 5 (solver.cu:764-799)
    {
      double t0 = wall_s();
      double nr = 0, bty = 0;
      const double* ax = s->h_out.p;
      double part[2 * kHostChunks] = {0};
      if (!s->dev_solve) host_ranges(m, [&](int c, int lo, int hi) {
        double a = 0, b = 0;
        for (int i = lo; i < hi; ++i) {
          const double rp = -ax[i] + s->b_p[i];
          s->Rp_p[i] = rp;
          const double ro = s->normA_p[i] * rp * s->bscale;
          a += ro * ro;
          b += s->b_p[i] * s->y_p[i];
        }
        part[2 * c] = a; part[2 * c + 1] = b;
      });
      for (int c = 0; c < kHostChunks; ++c) { nr += part[2 * c]; bty += part[2 * c + 1]; }
      {
        double v[4] = {nr, bty, s->h_out.p[(size_t)m], s->h_out.p[(size_t)m + 1]};
        if (from_batch) {
          for (int q = 0; q < 4; ++q) v[q] = s->bt.h.p[4 * (size_t)s->bt.pos + q];
          if (++s->bt.pos == s->bt.len) s->bt.len = s->bt.pos = 0;
        } else if (s->dev_scalars || s->dev_solve) { for (int q = 0; q < 4; ++q) v[q] = s->h_scal.p[q]; }   // formed (and summed over ranks) on the device
        else if ((rc = s->allreduce_scalars(v, 4))) return rc;
        nr = v[0]; bty = v[1]; s->h_out.p[(size_t)m] = v[2]; s->h_out.p[(size_t)m + 1] = v[3];
        // a lost row exchange of the four-workgroups-per-row tail kernel leaves NaN in y and with it in these sums: stop NOW (not at
        // max_iter), report once, and leave the handle on the two-GEMV path (TailSolve::take_failure)
        if (s->tail.d_fail && !(std::fabs(nr) <= 1.7976931348623157e308 && std::fabs(bty) <= 1.7976931348623157e308)) {
          const int tf = s->tail.take_failure(s->st);
          if (tf != 0) {
            set_error("solve: the dense tail of the A*A^T solve lost %d row exchanges at iteration %d (workgroups of a row not co-resident for seconds): "
                      "y is not valid; the handle now uses the two-pass tail kernels", tf, iter);
            return CUADMM_ERR_FACTOR;
          }
        }
      }
      s->errRp = std::sqrt(nr) / s->norm_borg;
      s->pobj = s->h_out.p[(size_t)m + 1] * s->objscale;
      s->errRd = std::sqrt(s->h_out.p[(size_t)m]) * s->Cscale / s->norm_Corg;
      s->dobj = bty * s->objscale;
      s->maxfeas = std::max(s->errRp, s->errRd);
      s->relgap = std::fabs(s->pobj - s->dobj) / (1 + std::fabs(s->pobj) + std::fabs(s->dobj));
      s->feasratio = s->ratioconst * s->errRp / s->errRd;
      if (s->feasratio < 1) s->prim_win += 1; else s->dual_win += 1;
      if ((iter <= sig_update_threshold && iter % sig_update_stage_1 == 1) ||
          (iter > sig_update_threshold && iter % sig_update_stage_2 == 1)) {
        if (s->prim_win > 1.2 * s->dual_win) { s->prim_win = 0; s->sig = std::min(s->sigmax, s->sig * sigscale); }
        else if (s->dual_win > 1.2 * s->prim_win) { s->dual_win = 0; s->sig = std::max(s->sigmin, s->sig / sigscale); }
      }
      s->prof_host(K_HOST, wall_s() - t0);
    }
    s->info[CUADMM_INFO_POBJ].push_back(s->pobj); s->info[CUADMM_INFO_DOBJ].push_back(s->dobj);
    s->info[CUADMM_INFO_ERRRP].push_back(s->errRp); s->info[CUADMM_INFO_ERRRD].push_back(s->errRd);
    s->info[CUADMM_INFO_RELGAP].push_back(s->relgap); s->info[CUADMM_INFO_SIG].push_back(s->sig);
    s->info[CUADMM_INFO_BSCALE].push_back(s->bscale); s->info[CUADMM_INFO_CSCALE].push_back(s->Cscale);
    s->info_iter_num++;
    // longest block first (PsdPlan::reorder_by_steps): at iterations 3, 24, 192, ... of this solve the stream is idle here
    if (s->fuse && lpt_ev > 0) ++s->lpt_iters;
    if (s->fuse && lpt_ev > 0 && s->lpt_iters >= lpt_ev && s->bt.len == 0 && s->steps_d.p && !s->can_batch()) {
      s->steps_h.resize(s->steps_d.n);
      if ((rc = staged_d2h(s->steps_h.data(), s->steps_d.p, sizeof(int) * s->steps_d.n, s->st))) return rc;
      if ((rc = s->plan.reorder_by_steps(s->steps_h.data(), s->st))) return rc;
      lpt_ev *= 8;
    }
  }

  // unscale (solver.cu:814-816) -- deferred until somebody reads or replaces X, y, S (materialise): a following
  // solve(if_first = false) continues from the scaled state.  With owned constraints over several ranks the y gather is a
  // collective, so there it happens here, where every rank is.
  s->pending_unscale = true;
  if (s->lazy_unscale == 0 || (s->local_mode && s->comm_world > 1)) { if ((rc = s->materialise())) return rc; }
  if (s->tail.k > 0) {
    const int tf = s->tail.take_failure(s->st);
    if (tf != 0) { set_error("solve: the dense tail of the A*A^T solve lost %d row exchanges (workgroups of a row not co-resident for seconds): y is not valid", tf); return CUADMM_ERR_FACTOR; }
  }
  s->eig_fail_total = s->plan.fail_count(s->st);
  if (s->eig_fail_total > 0) {
    set_error("solve: %d block projections hit the QL sweep cap", s->eig_fail_total);
    return CUADMM_ERR_EIG;
  }
  return CUADMM_OK;
}

int cuadmm_solver::materialise() {
  if (!pending_unscale) return CUADMM_OK;
  pending_unscale = false;
  int rc;
  if ((rc = check_device(device))) return rc;
  if ((rc = launch_scale(X.p, L, bscale, st))) return rc;
  if ((rc = launch_scale(S.p, L, Cscale, st))) return rc;
  if (dev_solve && m > 0) {
    if (y_registered) CUADMM_HIP_TRY(hipMemcpyAsync(y_p.data(), y_d.p, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, st));
    else {   // never a runtime copy into pageable memory (staging.hip)
      CUADMM_HIP_TRY(hipMemcpyAsync(h_y.p, y_d.p, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, st));
      CUADMM_HIP_TRY(hipStreamSynchronize(st));
      std::memcpy(y_p.data(), h_y.p, sizeof(double) * (size_t)m);
    }
  }
  CUADMM_HIP_TRY(hipStreamSynchronize(st));
  for (int i = 0; i < m; ++i) y_p[i] = y_p[i] / normA_p[i] * Cscale;
  if (local_mode) {   // y is replicated for the caller: gather the owned pieces once per solve
    y_full.assign((size_t)m_full, 0.0);
    for (int i = 0; i < m; ++i) y_full[cons_local[perm[i]]] = y_p[i];
    if (comm_world > 1 && m_full > 0) {
      if (!yfull_d.p && (rc = yfull_d.alloc((size_t)m_full))) return rc;
      if ((rc = staged_h2d(yfull_d.p, y_full.data(), sizeof(double) * (size_t)m_full, st))) return rc;
      if ((rc = comm_allreduce(yfull_d.p, (size_t)m_full))) return rc;
      if ((rc = staged_d2h(y_full.data(), yfull_d.p, sizeof(double) * (size_t)m_full, st))) return rc;
    }
  }
  return CUADMM_OK;
}

// SDPDuoSolver front (duo_solver.h:236-276): exactly two block sizes, then the generic engine.
int cuadmm_duo_init(cuadmm_solver* s, int if_gpu_eig_mom, int device_num_requested, int eig_stream_num_per_gpu,
                    int cpu_eig_thread_num, int vec_len, int con_num, const int* At_cp, const int* At_ri, const double* At_vx,
                    int At_nnz, const int* b_idx, const double* b_vals, int b_nnz, const int* C_idx, const double* C_vals,
                    int C_nnz, const int* blk, int mat_num, const double* X0, const double* y0, const double* S0, double sig) {
  if (!s || !blk || mat_num <= 0) { set_error("duo_init: invalid argument"); return CUADMM_ERR_INVALID; }
  // if_gpu_eig_mom = false asks for the reference's host-LAPACK moment-matrix path (duo_solver.cu:578-618,793-834).  This engine
  // has no CPU projection (and no CPU fallback by design): say so instead of silently running the GPU kernels; the option
  // "duo_cpu_eig_on_gpu" = 1 accepts the request and projects on the GPU (same iterates, to the projection's tolerance).
  if (!if_gpu_eig_mom && !s->duo_cpu_eig_on_gpu) {
    set_error("duo_init: if_gpu_eig_mom = false selects the reference's host LAPACK eigendecomposition; this engine projects on the GPU only "
              "(set option duo_cpu_eig_on_gpu = 1 to run the GPU projection for such a call)");
    return CUADMM_ERR_INVALID;
  }
  // duo_solver.cu:487-577 spreads the moment matrices over device_num_requested GPUs from ONE process (threads + P2P copies,
  // check_gpus.cu:29-43).  Here a GPU is a rank of the block-sharded engine: with rank / world already set by the caller (one
  // process per GPU) the request must agree with them; from a single process (world = 1) the handle becomes rank 0 of a GROUP of
  // N engines on N host threads with an in-process all-reduce (duo_group.hip).
  if (device_num_requested > 1 && s->world == 1 && !s->in_group_call) {
    if (s->group) { set_error("duo_init: this handle already leads a group of %d engines", duo_group_world(s->group)); return CUADMM_ERR_INVALID; }
    if (s->initialised) { set_error("duo_init: already initialised"); return CUADMM_ERR_INVALID; }
    DeviceGuard keep_device;                // the ranks' devices are made current on this thread: leave the caller's as it was
    int rc = duo_group_create(s, device_num_requested, s->device, s->duo_share_device != 0, s->duo_exchange, s->option_log, &s->group);
    if (rc) return rc;
    duo_group_inject(s->group, s->duo_inject);
    s->world = device_num_requested; s->rank = 0;
    rc = duo_group_run(s->group, [&](cuadmm_solver* q, int) {
      q->in_group_call = true;
      int r2 = cuadmm_duo_init(q, if_gpu_eig_mom, device_num_requested, eig_stream_num_per_gpu, cpu_eig_thread_num, vec_len, con_num, At_cp, At_ri,
                               At_vx, At_nnz, b_idx, b_vals, b_nnz, C_idx, C_vals, C_nnz, blk, mat_num, X0, y0, S0, sig);
      q->in_group_call = false;
      return r2;
    });
    if (rc) { duo_group_destroy(s->group); s->group = nullptr; s->world = 1; s->allreduce = nullptr; s->allreduce_user = nullptr; }
    return rc;
  }
  if (device_num_requested > 1 && device_num_requested != s->world) {
    set_error("duo_init: device_num_requested = %d but this handle is rank %d of %d: one process per GPU needs world = device_num_requested "
              "(or leave world = 1 and let duo_init build the in-process group)", device_num_requested, s->rank, s->world);
    return CUADMM_ERR_INVALID;
  }
  std::vector<int> sizes, nums;
  analyze_blk(blk, mat_num, sizes, nums);
  if (sizes.size() != 2) {   // analyze_blk_duo, src/utils/analyze_blk.cu:39-43
    set_error("SDP solver only supports two matrix sizes! You matrix type number is: %d", (int)sizes.size());
    return CUADMM_ERR_INVALID;
  }
  if (s->verbose && s->rank == 0) {
    std::cout << "\nAnalysis of the blk vector:" << std::endl;
    std::cout << "moment matrix size: " << sizes[1] << std::endl;
    std::cout << "localizing matrix size: " << sizes[0] << std::endl;
    std::cout << "number of moment matrices: " << nums[1] << std::endl;
    std::cout << "number of localizing matrices: " << nums[0] << std::endl << std::endl;
  }
  const int verbose = s->verbose;
  s->verbose = 0;   // the generic census is not part of the duo solver's console output
  int rc = cuadmm_init(s, eig_stream_num_per_gpu, cpu_eig_thread_num, vec_len, con_num, At_cp, At_ri, At_vx, At_nnz, b_idx, b_vals,
                       b_nnz, C_idx, C_vals, C_nnz, blk, mat_num, X0, y0, S0, sig);
  s->verbose = verbose;
  return rc;
}
int cuadmm_duo_solve(cuadmm_solver* s, int max_iter, double stop_tol, int sig_update_threshold, int sig_update_stage_1,
                     int sig_update_stage_2, int switch_admm, double sigscale, int if_first) {
  return cuadmm_solve(s, max_iter, stop_tol, sig_update_threshold, sig_update_stage_1, sig_update_stage_2, switch_admm, sigscale, if_first);
}

int cuadmm_get_dims(const cuadmm_solver* s, int* vec_len, int* con_num, int* mat_num) {
  if (!s) { set_error("get_dims: null"); return CUADMM_ERR_INVALID; }
  // the caller's numbering in every sharding mode: cuadmm_get_y writes con_num doubles, cuadmm_set_XyS / cuadmm_get_shard
  // index the global svec
  if (vec_len) *vec_len = s->local_mode ? s->L_caller : s->L_full;
  if (con_num) *con_num = s->local_mode ? s->m_full : s->m;
  if (mat_num) *mat_num = s->local_mode ? s->nblk_caller : s->nblk_full;
  return CUADMM_OK;
}

static int get_vec(cuadmm_solver* s, const DevBuf<double>& v, double* out) {
  if (!s || !s->initialised || !out) { set_error("get: bad arguments"); return CUADMM_ERR_INVALID; }
  int rc = check_device(s->device);
  if (rc) return rc;
  if ((rc = s->materialise())) return rc;
  CUADMM_HIP_TRY(hipStreamSynchronize(s->st));
  if (s->L > 0) { int rc2 = staged_d2h(out, v.p, sizeof(double) * (size_t)s->L, s->st); if (rc2) return rc2; }
  return CUADMM_OK;
}
// world>1: writes this rank's shard (length svec_end - svec_begin) at out[0..)
// the leader of an in-process group gathers the shards: out has the caller's vec_len doubles
static int group_gather(cuadmm_solver* s, double* out, int (*get)(cuadmm_solver*, double*)) {
  DeviceGuard keep_device;                  // every rank's getter makes its device current on this thread
  for (int r = 0; r < duo_group_world(s->group); ++r) {
    cuadmm_solver* q = duo_group_rank(s->group, r);
    int64_t b = 0, e = 0;
    q->in_group_call = true;                // rank 0 IS the leader: without the mark get_shard would answer for the whole group
    int rc = cuadmm_get_shard(q, &b, &e, nullptr, nullptr);
    if (!rc) rc = e > b ? get(q, out + b) : CUADMM_OK;
    q->in_group_call = false;
    if (rc) return rc;
  }
  return CUADMM_OK;
}
int cuadmm_get_X(cuadmm_solver* s, double* out) {
  if (s && s->group && !s->in_group_call) return group_gather(s, out, cuadmm_get_X);
  return get_vec(s, s->X, out);
}
int cuadmm_get_S(cuadmm_solver* s, double* out) {
  if (s && s->group && !s->in_group_call) return group_gather(s, out, cuadmm_get_S);
  return get_vec(s, s->S, out);
}
int cuadmm_get_y(cuadmm_solver* s, double* out) {
  if (!s || !s->initialised || !out) { set_error("get_y: bad arguments"); return CUADMM_ERR_INVALID; }
  { int rc = s->materialise(); if (rc) return rc; }
  if (s->local_mode) {
    if ((int)s->y_full.size() == s->m_full) std::copy(s->y_full.begin(), s->y_full.end(), out);   // gathered at the end of solve
    else { std::fill(out, out + s->m_full, 0.0); for (int i = 0; i < s->m; ++i) out[s->cons_local[s->perm[i]]] = s->y_p[i]; }
    return CUADMM_OK;
  }
  for (int i = 0; i < s->m; ++i) out[s->perm[i]] = s->y_p[i];          // y[perm[i]] = y_perm[i], solver.cu:500
  return CUADMM_OK;
}

int cuadmm_set_XyS(cuadmm_solver* s, const double* X, const double* y, const double* S, double sig) {
  if (!s || !s->initialised) { set_error("set_XyS: not initialised"); return CUADMM_ERR_INVALID; }
  if (s->group && !s->in_group_call) {      // every rank of the group takes its range of the caller's vectors
    DeviceGuard keep_device;
    for (int r = 0; r < duo_group_world(s->group); ++r) {
      cuadmm_solver* q = duo_group_rank(s->group, r);
      q->in_group_call = true;
      int rc = cuadmm_set_XyS(q, X, y, S, sig);
      q->in_group_call = false;
      if (rc) return rc;
    }
    return CUADMM_OK;
  }
  int rc = check_device(s->device);
  if (rc) return rc;
  if ((rc = s->materialise())) return rc;       // the vectors NOT replaced must be in the caller's units too
  if (X && s->L > 0) { int rc2 = staged_h2d(s->X.p, X + s->sv_off + s->sv_begin, sizeof(double) * (size_t)s->L, s->st); if (rc2) return rc2; }
  if (S && s->L > 0) { int rc2 = staged_h2d(s->S.p, S + s->sv_off + s->sv_begin, sizeof(double) * (size_t)s->L, s->st); if (rc2) return rc2; }
  if (y) for (int i = 0; i < s->m; ++i) s->y_p[i] = s->local_mode ? y[s->cons_local[s->perm[i]]] : y[s->perm[i]];
  if (sig > 0) s->sig = sig;
  return CUADMM_OK;
}

int cuadmm_get_device_ptrs(cuadmm_solver* s, double** X, double** y, double** S) {
  if (!s || !s->initialised) { set_error("get_device_ptrs: not initialised"); return CUADMM_ERR_INVALID; }
  { int rc = s->materialise(); if (rc) return rc; }
  if (X) *X = s->X.p;
  if (y) *y = s->y_d.p;
  if (S) *S = s->S.p;
  return CUADMM_OK;
}

int cuadmm_get_shard(const cuadmm_solver* s, int64_t* b, int64_t* e, int* kb, int* ke) {
  if (!s || !s->initialised) { set_error("get_shard: not initialised"); return CUADMM_ERR_INVALID; }
  if (s->group && !s->in_group_call) {      // the leader's getters return whole vectors
    if (b) *b = 0;
    if (e) *e = s->local_mode ? s->L_caller : s->L_full;
    if (kb) *kb = 0;
    if (ke) *ke = s->local_mode ? s->nblk_caller : s->nblk_full;
    return CUADMM_OK;
  }
  if (b) *b = s->sv_off + s->sv_begin;
  if (e) *e = s->sv_off + s->sv_end;
  if (kb) *kb = s->blk_off + s->blk_begin;
  if (ke) *ke = s->blk_off + s->blk_end;
  return CUADMM_OK;
}

int cuadmm_get_info_iter_num(const cuadmm_solver* s) { return s ? s->info_iter_num : 0; }
int cuadmm_get_info_array(const cuadmm_solver* s, int which, double* out, int cap) {
  if (!s || which < 0 || which >= 8 || !out) { set_error("get_info_array: bad arguments"); return CUADMM_ERR_INVALID; }
  // the reference appends across solve() calls (vectors are never cleared, solver.cu:802-809)
  int n = std::min<int>(cap, (int)s->info[which].size());
  std::copy(s->info[which].begin(), s->info[which].begin() + n, out);
  return n;
}
double cuadmm_get_total_time(const cuadmm_solver* s) { return s ? s->total_time : 0.0; }
int cuadmm_get_state(const cuadmm_solver* s, double o[12]) {
  if (!s || !o) { set_error("get_state: null"); return CUADMM_ERR_INVALID; }
  o[0] = s->errRp; o[1] = s->errRd; o[2] = s->pobj; o[3] = s->dobj; o[4] = s->relgap; o[5] = s->sig;
  o[6] = s->bscale; o[7] = s->Cscale; o[8] = s->norm_borg; o[9] = s->norm_Corg; o[10] = s->best_KKT;
  o[11] = (double)s->eig_fail_total;
  return CUADMM_OK;
}

int cuadmm_get_profile(const cuadmm_solver* s, double out[3 * CUADMM_NUM_KCLASS]) {
  if (!s || !out) { set_error("get_profile: null"); return CUADMM_ERR_INVALID; }
  for (int k = 0; k < K_NUM; ++k) { out[3 * k] = s->prof_count[k]; out[3 * k + 1] = s->prof_ms[k]; out[3 * k + 2] = s->prof_bytes[k]; }
  return CUADMM_OK;
}
int cuadmm_get_psd_steps(cuadmm_solver* s, int* out, int cap) {
  if (!s || !out) { set_error("get_psd_steps: null"); return CUADMM_ERR_INVALID; }
  if (!s->steps_d.p && s->psd_steps && s->initialised && s->blk_local.empty()) return 0;     // a rank without blocks (PlanarHand_N=1 on eight ranks: its
                                                                                            // n = 120 block alone outweighs a rank's share) has nothing to report
  if (!s->steps_d.p) { set_error("get_psd_steps: set option psd_steps=1 before init"); return CUADMM_ERR_INVALID; }
  const int n = (int)std::min<size_t>(s->steps_d.n, (size_t)std::max(cap, 0));
  CUADMM_HIP_TRY(hipStreamSynchronize(s->st));
  { int rc2 = staged_d2h(out, s->steps_d.p, sizeof(int) * (size_t)n, s->st); if (rc2) return rc2; }
  return n;
}
int cuadmm_get_counters(const cuadmm_solver* s, double o[8]) {
  if (!s || !o) { set_error("get_counters: null"); return CUADMM_ERR_INVALID; }
  o[0] = (double)s->bt.launches; o[1] = (double)s->bt.iters; o[2] = (double)s->bt.rollbacks; o[3] = (double)cuadmm_host_pool_threads();
  o[4] = s->fuse ? 1 : 0; o[5] = s->closed.active ? 1 : 0; o[6] = s->dev_solve ? (s->lead.tops ? 3 : 1) : (s->lead.hybrid ? 2 : 0); o[7] = (double)s->tail.k;
  return CUADMM_OK;
}
int cuadmm_get_tail_info(const cuadmm_solver* s, double o[6]) {
  if (!s || !o) { set_error("get_tail_info: null"); return CUADMM_ERR_INVALID; }
  o[0] = (double)s->tail.k; o[1] = s->tail.shard_bytes; o[2] = (double)s->tail.shard_rows; o[3] = s->tail.resident_bytes;
  o[4] = s->tail.inv_resid; o[5] = (s->tail.refine && s->tail.Lm) ? 1.0 : 0.0;
  return CUADMM_OK;
}
int cuadmm_get_group_info(const cuadmm_solver* s, double o[4]) {
  if (!s || !o) { set_error("get_group_info: null"); return CUADMM_ERR_INVALID; }
  o[0] = s->group ? (double)duo_group_world(s->group) : 1.0;
  o[1] = s->group ? (double)duo_group_exchange(s->group) : 0.0;
  o[2] = s->group ? (double)duo_group_distinct_devices(s->group) : 1.0;
  o[3] = s->group ? (double)duo_group_allreduces(s->group) : 0.0;
  return CUADMM_OK;
}
int cuadmm_reset_profile(cuadmm_solver* s) {
  if (!s) { set_error("reset_profile: null"); return CUADMM_ERR_INVALID; }
  for (int k = 0; k < K_NUM; ++k) { s->prof_count[k] = 0; s->prof_ms[k] = 0; }
  return CUADMM_OK;
}

// ------------------------------------------------------------------------------------------
// op-level entry points that need the planner
// ------------------------------------------------------------------------------------------
int cuadmm_op_batch_eig(double* mat, double* W, int* info, int n, int count, void* stream) {
  return psd_batch_eig(mat, W, info, n, count, (hipStream_t)stream);
}

int cuadmm_op_gemm_sym(int n, const double* A, const double* B, double alpha, double beta, const double* E, double* Cout, void* stream) {
  if (n < 64 || n % 64 != 0 || !A || !B || !Cout) { set_error("gemm_sym: n must be a positive multiple of 64 and pointers non-null"); return CUADMM_ERR_INVALID; }
  return large_gemm_sym(n, A, B, alpha, beta, E, Cout, (hipStream_t)stream);
}

int cuadmm_op_tail_factor_solve(const int64_t* row_ptr, const int* col, const double* val, int k, double* z_host, int nrhs) {
  if (!row_ptr || !col || !val || !z_host || k < 1 || nrhs < 0) { set_error("tail_factor_solve: bad arguments"); return CUADMM_ERR_INVALID; }
  TailSolve t;
  int rc = t.build_from_schur(reinterpret_cast<const long long*>(row_ptr), col, val, k, nullptr);
  for (int r = 0; r < nrhs && !rc; ++r) rc = t.solve(z_host + (size_t)r * k, nullptr);
  return rc;
}

int cuadmm_op_tail_solve(const double* L22_host, const double* D2_host, int k, double* z2_host, int nrhs) {
  if (!L22_host || !D2_host || !z2_host || k < 1 || nrhs < 0) { set_error("tail_solve: bad arguments"); return CUADMM_ERR_INVALID; }
  TailSolve t;
  int rc = t.build(L22_host, D2_host, k, nullptr);
  for (int r = 0; r < nrhs && !rc; ++r) rc = t.solve(z2_host + (size_t)r * k, nullptr);
  if (!rc) {
    const int lost = t.fail_count(nullptr);
    if (lost != 0) { set_error("tail_solve: %d row exchanges lost (workgroups of a row not co-resident)", lost); return CUADMM_ERR_FACTOR; }
  }
  return rc;
}

int cuadmm_op_tail_solve_sharded(const double* L22_host, const double* D2_host, int k, const double* z_host, int world, int one_pass,
                                 double* out_host, int* rows_out) {
  if (!L22_host || !D2_host || !z_host || !out_host || k < 1 || world < 1) { set_error("tail_solve_sharded: bad arguments"); return CUADMM_ERR_INVALID; }
  const bool compact = (one_pass & 2) != 0;      // + 2: every rank KEEPS its rows only (TailSolve::keep_shard), one object per rank
  int rc = CUADMM_OK;
  for (int p = 0; p < world && !rc; ++p) {
    TailSolve t;
    t.one_pass = (one_pass & 1) != 0;
    if ((rc = t.build(L22_host, D2_host, k, nullptr))) break;
    // the reduction is left out: every rank's partial result comes back as it stands
    t.reduce_fn = [](void*, double*, size_t, hipStream_t) -> int { return CUADMM_OK; };
    t.shard_world = world;
    const int p_end = compact ? p + 1 : world;   // without compaction ONE object serves every rank in turn
    for (int q = p; q < p_end && !rc; ++q) {
      t.shard_rank = q;
      if (compact && (rc = t.keep_shard(q, world, nullptr))) break;
      std::copy(z_host, z_host + k, out_host + (size_t)q * k);
      rc = t.solve(out_host + (size_t)q * k, nullptr);
      if (rows_out) rows_out[q] = t.shard_rows;
    }
    if (!rc) {
      const int lost = t.fail_count(nullptr);
      if (lost != 0) { set_error("tail_solve_sharded: %d row exchanges lost", lost); return CUADMM_ERR_FACTOR; }
    }
    if (!compact) break;
  }
  return rc;
}

// Failure drill of the four-workgroups-per-row tail kernel (18 432 < K <= 32 768), see the protocol above ts_onepass_group_kernel:
// out[0] = the plain solve of z; out[1] = the solve of z with a NaN carrying the exchange sentinel's bits in z[0] (must come back as
// NaN promptly, no exchange lost: counts[0] = 0); out[2] = the solve with the failure counter raised beforehand (NaN at once,
// counts[1] = the count take_failure() found and cleared); out[3] = the solve after that, on the two-GEMV path the object retired to
// (counts[2] = 1).  Test hook only.
int cuadmm_op_tail_solve_drill(const double* L22_host, const double* D2_host, int k, const double* z_host, double* out_host, int* counts) {
  if (!L22_host || !D2_host || !z_host || !out_host || !counts || k < 1) { set_error("tail_solve_drill: bad arguments"); return CUADMM_ERR_INVALID; }
  TailSolve t;
  int rc = t.build(L22_host, D2_host, k, nullptr);
  if (rc) return rc;
  if (!t.d_fail) { set_error("tail_solve_drill: k = %d does not use the row-sharing kernel", k); return CUADMM_ERR_INVALID; }
  for (int q = 0; q < 4; ++q) std::copy(z_host, z_host + k, out_host + (size_t)q * k);
  if ((rc = t.solve(out_host, nullptr))) return rc;
  { const unsigned long long bits = 0x7ff8dead5eed0001ull; std::memcpy(out_host + (size_t)k, &bits, sizeof bits); }
  if ((rc = t.solve(out_host + (size_t)k, nullptr))) return rc;
  counts[0] = t.fail_count(nullptr);
  CUADMM_HIP_TRY(hipMemset(t.d_fail, 0, sizeof(int)));
  { const int one = 1; CUADMM_HIP_TRY(hipMemcpy(t.d_fail, &one, sizeof one, hipMemcpyHostToDevice)); }
  if ((rc = t.solve(out_host + 2 * (size_t)k, nullptr))) return rc;
  counts[1] = t.take_failure(nullptr);
  counts[2] = t.group_retired ? 1 : 0;
  return t.solve(out_host + 3 * (size_t)k, nullptr);
}

int cuadmm_op_psd_project(const double* Xb, double* Xproj, const int* blk_host, int mat_num, void* stream) {
  return cuadmm_op_psd_project_steps(Xb, Xproj, blk_host, mat_num, nullptr, stream);
}

int cuadmm_op_psd_project_steps(const double* Xb, double* Xproj, const int* blk_host, int mat_num, int* steps_dev, void* stream) {
  return cuadmm_op_psd_project_ex(Xb, Xproj, blk_host, mat_num, 0, steps_dev, stream);
}

int cuadmm_op_psd_project_ex(const double* Xb, double* Xproj, const int* blk_host, int mat_num, int eig_rank, int* steps_dev, void* stream) {
  if (!blk_host || mat_num < 0 || eig_rank < 0) { set_error("psd_project: bad arguments"); return CUADMM_ERR_INVALID; }
  PsdPlan plan;
  plan.eig_rank = eig_rank;
  int rc = plan.build(blk_host, mat_num);
  if (rc) return rc;
  if (steps_dev) {
    CUADMM_HIP_TRY(hipMemsetAsync(steps_dev, 0, sizeof(int) * (size_t)mat_num, (hipStream_t)stream));
    plan.d_steps = steps_dev;
    plan.sign.d_steps = steps_dev;
  }
  rc = plan.project(Xb, Xproj, (hipStream_t)stream);
  if (rc) return rc;
  int fails = plan.fail_count((hipStream_t)stream);   // synchronises the stream
  if (fails != 0) { set_error("psd_project: %d blocks hit the QL sweep cap", fails); return CUADMM_ERR_EIG; }
  return CUADMM_OK;
}

struct cuadmm_psd_plan { PsdPlan plan; };
int cuadmm_psd_plan_create(const int* blk_host, int mat_num, int eig_rank, cuadmm_psd_plan** out) {
  if (!blk_host || mat_num < 0 || eig_rank < 0 || !out) { set_error("psd_plan_create: bad arguments"); return CUADMM_ERR_INVALID; }
  cuadmm_psd_plan* p = new cuadmm_psd_plan();
  p->plan.eig_rank = eig_rank;
  int rc = p->plan.build(blk_host, mat_num);
  if (rc) { delete p; return rc; }
  p->plan.overlap = true;
  *out = p;
  return CUADMM_OK;
}
int cuadmm_psd_plan_project(cuadmm_psd_plan* p, const double* Xb, double* Xproj, int* steps_dev, void* stream) {
  if (!p || !Xb || !Xproj) { set_error("psd_plan_project: bad arguments"); return CUADMM_ERR_INVALID; }
  p->plan.d_steps = steps_dev;
  p->plan.sign.d_steps = steps_dev;
  return p->plan.project(Xb, Xproj, (hipStream_t)stream);
}
void cuadmm_psd_plan_destroy(cuadmm_psd_plan* p) { delete p; }

int cuadmm_dev_malloc(void** ptr, size_t bytes) { CUADMM_HIP_TRY(hipMalloc(ptr, bytes ? bytes : 8)); return CUADMM_OK; }
int cuadmm_dev_free(void* ptr) { if (ptr) CUADMM_HIP_TRY(hipFree(ptr)); return CUADMM_OK; }
int cuadmm_memcpy_h2d(void* dst, const void* src, size_t bytes) { return staged_h2d(dst, src, bytes); }
int cuadmm_memcpy_d2h(void* dst, const void* src, size_t bytes) { return staged_d2h(dst, src, bytes); }
int cuadmm_dev_sync(void) { CUADMM_HIP_TRY(hipDeviceSynchronize()); return CUADMM_OK; }

// ------------------------------------------------------------------------------------------
// TXT problem objects
// ------------------------------------------------------------------------------------------
struct cuadmm_problem { ProblemData d; };

int cuadmm_problem_from_txt(const char* prefix, cuadmm_problem** out) {
  if (!prefix || !out) { set_error("problem_from_txt: null"); return CUADMM_ERR_INVALID; }
  cuadmm_problem* p = new cuadmm_problem();
  int rc = load_problem_txt(prefix, p->d, true);
  if (rc) { delete p; return rc; }
  *out = p;
  return CUADMM_OK;
}
int cuadmm_problem_view_get(const cuadmm_problem* p, cuadmm_problem_view* v) {
  if (!p || !v) { set_error("problem_view: null"); return CUADMM_ERR_INVALID; }
  const ProblemData& d = p->d;
  v->vec_len = d.vec_len; v->con_num = d.con_num; v->mat_num = d.mat_num;
  v->At_nnz = (int)d.At_vals.size(); v->b_nnz = (int)d.b_vals.size(); v->C_nnz = (int)d.C_vals.size();
  v->At_csc_col_ptrs = d.At_col_ptrs.data(); v->At_csc_row_ids = d.At_row_ids.data(); v->At_csc_vals = d.At_vals.data();
  v->b_indices = d.b_idx.data(); v->b_vals = d.b_vals.data(); v->C_indices = d.C_idx.data(); v->C_vals = d.C_vals.data();
  v->blk_vals = d.blk.data();
  return CUADMM_OK;
}
void cuadmm_problem_free(cuadmm_problem* p) { delete p; }

int cuadmm_coo_to_csc(int* col_ptrs, int* col_ids, int* row_ids, double* vals, int nnz, int col_num) {
  if (nnz < 0 || col_num < 0 || !col_ptrs) { set_error("coo_to_csc: bad arguments"); return CUADMM_ERR_INVALID; }
  std::vector<int> cp, c(col_ids, col_ids + nnz), r(row_ids, row_ids + nnz);
  std::vector<double> v(vals, vals + nnz);
  coo_to_csc(cp, c, r, v, nnz, col_num);
  std::copy(cp.begin(), cp.end(), col_ptrs);
  std::copy(c.begin(), c.end(), col_ids);
  std::copy(r.begin(), r.end(), row_ids);
  std::copy(v.begin(), v.end(), vals);
  return CUADMM_OK;
}

int cuadmm_read_blk(const char* filename, char* types, int* sizes, int cap) {
  std::vector<char> t;
  std::vector<int> sz;
  int rc = read_blk_file(filename, t, sz);
  if (rc) return rc;
  for (size_t i = 0; i < sz.size() && (int)i < cap; ++i) { if (types) types[i] = t[i]; if (sizes) sizes[i] = sz[i]; }
  return (int)sz.size();
}

int cuadmm_write_dense_txt(const char* filename, const double* vals, int64_t n) {
  FILE* f = fopen(filename, "w");
  if (!f) { set_error("Failed to open file: %s", filename); return CUADMM_ERR_IO; }
  for (int64_t i = 0; i < n; ++i) fprintf(f, "%.32f\n", vals[i]);     // memory.h:278-294
  fclose(f);
  return CUADMM_OK;
}

}  // extern "C"
