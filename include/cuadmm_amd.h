/*
 * cuadmm_amd.h -- C ABI of the MI355X-native SDP-ADMM iteration engine.
 *
 * Drop-in boundary for the hot path of ComputationalRobotics/cuADMM: the class
 * SDPSolver (reference include/cuadmm/solver.h:30-248, src/solver.cu) as driven by its two
 * front ends (src/main.cu:22-41, MATLAB/cuadmm_MATLAB.cu:342-424).  The reference has no
 * FFI layer; a maintainer binds these entry points where the reference constructs and
 * calls SDPSolver.  Plain pointers and sizes only; all indices are
 * 0-based int32 as in the reference; all vectors fp64.
 *
 * Every function returns CUADMM_OK (0) or a negative error code; cuadmm_last_error()
 * returns a human readable message for the calling thread.  Nothing here falls back to a
 * CPU implementation of the device path: without a usable gfx950 device the compute entry
 * points fail with CUADMM_ERR_NO_DEVICE.
 */
#ifndef CUADMM_AMD_H
#define CUADMM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CUADMM_OK 0
#define CUADMM_ERR_INVALID (-1)    /* bad argument / bad state                          */
#define CUADMM_ERR_NO_DEVICE (-2)  /* no HIP device, or a HIP call failed               */
#define CUADMM_ERR_IO (-3)         /* unreadable / malformed input file                 */
#define CUADMM_ERR_FACTOR (-4)     /* A A^T factorisation failed                        */
#define CUADMM_ERR_EIG (-5)        /* an eigen-iteration hit its iteration cap          */
#define CUADMM_ERR_COMM (-6)       /* collective hook failed                            */
#define CUADMM_ERR_ALLOC (-7)      /* out of host memory (std::bad_alloc caught at the boundary) */
#define CUADMM_ERR_INTERNAL (-8)   /* a C++ exception other than bad_alloc reached the boundary  */

const char* cuadmm_last_error(void);
const char* cuadmm_version(void);
/* number of visible HIP devices (0 if none); never initialises a device context */
int cuadmm_device_count(void);

/* ------------------------------------------------------------------------------------ */
/* Solver object: replaces `SDPSolver solver;` (reference src/main.cu:21).               */
/* ------------------------------------------------------------------------------------ */
typedef struct cuadmm_solver cuadmm_solver;

int cuadmm_create(cuadmm_solver** out);
void cuadmm_destroy(cuadmm_solver* s);

/* Engine options; call before cuadmm_init.  Unknown keys -> CUADMM_ERR_INVALID.
 *   "device"        HIP device ordinal (default 0; the reference hard-wires GPU0, utils.h:4)
 *   "verbose"       1 = print the reference's census + iteration table to stdout (default 1)
 *   "rank","world"  shard blocks by index over `world` engines (default 0,1); see
 *                   cuadmm_set_allreduce
 *   "profile"       1 = time every kernel class with HIP events on the engine stream;
 *                   2 = time only the dominant kernel (psd_project)
 *   "force_comm"    1 = call the collective hook even when world == 1 (transport tests on one GPU)
 *   "eig_rank"      r > 0: rank-limited projection (only the r largest eigenvalues of every PSD block survive; reference
 *                   dense_scalar.cu:51-57 + get_eig_rank_mask.cu, dormant there); active from iteration
 *                   "eig_rank_begin_iter" (default 0) on, or once maxfeas < "eig_rank_maxfeas" (default 0 = never);
 *                   set before cuadmm_init (every block then takes the eigensolver kernels)
 *   "psd_steps"     1 = record how many Newton-Schulz steps the adaptive matrix-sign projection took per block
 *                   (cuadmm_get_psd_steps); set before cuadmm_init
 *   "graph"         reserved
 *   "tail_shard"    world > 1, coupled constraints: 1 (default) = each rank applies 1 / world of the rows of the dense GPU tail of the
 *                   replicated y-solve and the K partial results are all-reduced; 0 = every rank applies the whole tail
 *   "tail_pivot"    1 (default) = the dense LDL^T of the y-solve's GPU tail pivots on the diagonal (P S P^T = L D L^T, |L_ij| <= 1): its explicit inverse stays
 *                   accurate where the Schur complement is nearly singular; 0 = the unpivoted elimination order (rounds 2 - 5)
 *   "tail_refine"   1 = one more accuracy device of the dense GPU tail of the y-solve: one refinement step of each triangular solve against the factor itself
 *                   (u <- u + W (z - L u), x <- x + W^T (v - L^T x)) on top of the explicit inverse W = inv(L22), whose accuracy is u cond(L22) -- a
 *                   nearly singular Schur complement (large moment relaxations) otherwise leaves 1e-8 ... 1e-7 in the primal objective against an exact
 *                   LDL^T solve (the reference's contract, include/cuadmm/cholesky_cpu.h:146-155).  Costs 6x the tail's bytes per solve and two more
 *                   K x K matrices; default 0.  cuadmm_get_tail_info [4] reports the measured accuracy of the explicit inverse.
 *   "tail_order", "tail_zreg", "tail_depth", "tail_rb"   the tail's one pass over inv(L22) (csrc/tail_solve.hip): the workgroups' row walk (2, default:
 *                   alternately from the long and the short end, odd workgroups starting short; 0 = longest first), z in registers (1) or LDS (0), row groups in
 *                   flight beyond the current one (0 ... 3, default 1), rows per barrier (0 = by size).  A/B switches: the same solve to the last few bits.
 *   "psd_lg_fuse"   1 (default) = a handful of blocks of 65 <= n <= 512 run their whole projection -- svec -> dense, norm, every step of the matrix-sign
 *                   iteration, final product, svec store -- in ONE launch; 0 = prologue and epilogue as launches of their own (bit-identical)
 *   "psd_lg_clean"  1 = the uncapped ("clean") mega-lift of the sign schedule for groups padded to <= 480 (csrc/sign_sched.h); default 0: measured slower
 *                   on every shipped input
 *   "duo_share_device", "duo_exchange"   the in-process group of cuadmm_duo_init(device_num_requested = N): all engines on the
 *                   caller's device; all-reduce through device memory (1), host staging (0), chosen by peer accessibility (-1, default)
 *   (every other switch: INTEGRATION.md section 6)
 */
int cuadmm_set_option(cuadmm_solver* s, const char* key, double value);

/* Collective hook for block-sharded multi-GPU runs.  `fn` must sum
 * `count` doubles at device pointer `buf` over all ranks, in place, ordered on `hip_stream`
 * (a hipStream_t).  bench.py installs torch.distributed (RCCL) here; cuadmm_use_rccl
 * installs a direct RCCL communicator instead. */
typedef int (*cuadmm_allreduce_fn)(void* user, double* buf, size_t count, void* hip_stream);
int cuadmm_set_allreduce(cuadmm_solver* s, cuadmm_allreduce_fn fn, void* user);
/* Direct RCCL: `unique_id` is the 128-byte ncclUniqueId shared by all ranks
 * (cuadmm_rccl_unique_id fills one on rank 0). */
int cuadmm_rccl_unique_id(char out128[128]);
int cuadmm_use_rccl(cuadmm_solver* s, const char unique_id128[128], int rank, int world);

/* Block types: blk_vals[k] > 0 is a PSD block of that size ('s n' in blk.txt); blk_vals[k] < 0 is an UNCONSTRAINED block of
 * -blk_vals[k] variables ('u n', reference README.md:55-64, "WIP" there: its loader rejects it, problem.cu:28-36), which owns
 * that many svec slots and is left untouched by the cone projection.  cuadmm_problem_from_txt maps 'u n' lines to -n. */
/* SDPSolver::init  (reference include/cuadmm/solver.h:208-223, src/solver.cu:27-342).
 * Same argument list and meaning.  At is the CSC of A^T (vec_len x con_num): column j =
 * constraint j, row ids = svec indices.  X/y/S may be NULL (cold start, zeros).  Caller keeps
 * ownership of every array; they are copied.  `eig_stream_num_per_gpu` and
 * `cpu_eig_thread_num` are accepted for signature compatibility (the reference ignores the
 * latter too, solver.cu:29). */
int cuadmm_init(cuadmm_solver* s,
                int eig_stream_num_per_gpu, int cpu_eig_thread_num,
                int vec_len, int con_num,
                const int* At_csc_col_ptrs, const int* At_csc_row_ids, const double* At_csc_vals, int At_nnz,
                const int* b_indices, const double* b_vals, int b_nnz,
                const int* C_indices, const double* C_vals, int C_nnz,
                const int* blk_vals, int mat_num,
                const double* X, const double* y, const double* S,
                double sig);

/* SDPSolver::solve  (reference solver.h:236-244, src/solver.cu:355-823).  Reference
 * defaults: sig_update_threshold=500, stage_1=50, stage_2=100, switch_admm=11000,
 * sigscale=1.05, if_first=1. */
int cuadmm_solve(cuadmm_solver* s, int max_iter, double stop_tol,
                 int sig_update_threshold, int sig_update_stage_1, int sig_update_stage_2,
                 int switch_admm, double sigscale, int if_first);

/* SDPDuoSolver::init / ::solve  (reference include/cuadmm/duo_solver.h:236-276, src/duo_solver.cu): the
 * two-block-size specialisation (moment + localizing matrices).  Same iteration; the reference differs only in
 * where the moment-matrix eigendecompositions run (`if_gpu_eig_mom`: N GPUs through threads + P2P copies, or
 * `cpu_eig_thread_num` host LAPACK threads).  Here every block is projected by the kernels of this engine:
 *   if_gpu_eig_mom = 0 (host-LAPACK moment matrices) is REFUSED with CUADMM_ERR_INVALID unless option
 *     "duo_cpu_eig_on_gpu" = 1 says that running them on the GPU instead is what the caller wants;
 *   device_num_requested = N > 1 from ONE process runs N engines on N host threads (devices 0 .. N-1, or all on
 *     this solver's device with option "duo_share_device" = 1), blocks sharded by index, the exchange step an
 *     in-process all-reduce through peer-visible staging buffers; with rank / world already
 *     set by the caller (one process per GPU) N must equal world.
 * Like the reference (analyze_blk.cu:39-43) init rejects inputs that do not have exactly two distinct block sizes. */
int cuadmm_duo_init(cuadmm_solver* s,
                    int if_gpu_eig_mom, int device_num_requested,
                    int eig_stream_num_per_gpu, int cpu_eig_thread_num,
                    int vec_len, int con_num,
                    const int* At_csc_col_ptrs, const int* At_csc_row_ids, const double* At_csc_vals, int At_nnz,
                    const int* b_indices, const double* b_vals, int b_nnz,
                    const int* C_indices, const double* C_vals, int C_nnz,
                    const int* blk_vals, int mat_num,
                    const double* X, const double* y, const double* S,
                    double sig /* reference default 2e2 */);
int cuadmm_duo_solve(cuadmm_solver* s, int max_iter, double stop_tol,
                     int sig_update_threshold, int sig_update_stage_1, int sig_update_stage_2,
                     int switch_admm, double sigscale, int if_first);

/* Results: the reference exposes public members X.vals / y.vals / S.vals (device pointers,
 * unscaled after solve, solver.cu:814-816) and info_* vectors (solver.h:149-161). */
int cuadmm_get_dims(const cuadmm_solver* s, int* vec_len, int* con_num, int* mat_num);
int cuadmm_get_X(cuadmm_solver* s, double* host_out /* vec_len */);
int cuadmm_get_y(cuadmm_solver* s, double* host_out /* con_num */);
int cuadmm_get_S(cuadmm_solver* s, double* host_out /* vec_len */);
/* replace the (unscaled) iterate between two solve() calls, as the reference allows by
 * writing through X.vals/y.vals/S.vals before solve(..., if_first=false) (solver.cu:385-409);
 * NULL leaves that vector untouched */
int cuadmm_set_XyS(cuadmm_solver* s, const double* X, const double* y, const double* S, double sig);
/* device pointers of this rank's shard (scaled inside solve, unscaled after) */
int cuadmm_get_device_ptrs(cuadmm_solver* s, double** X, double** y, double** S);
/* svec range [begin,end) owned by this rank (whole vector when world==1) */
int cuadmm_get_shard(const cuadmm_solver* s, int64_t* svec_begin, int64_t* svec_end, int* blk_begin, int* blk_end);

#define CUADMM_INFO_POBJ 0
#define CUADMM_INFO_DOBJ 1
#define CUADMM_INFO_ERRRP 2
#define CUADMM_INFO_ERRRD 3
#define CUADMM_INFO_RELGAP 4
#define CUADMM_INFO_SIG 5
#define CUADMM_INFO_BSCALE 6
#define CUADMM_INFO_CSCALE 7
int cuadmm_get_info_iter_num(const cuadmm_solver* s);
/* copies min(cap, iter_num) entries of info_<which>_arr; returns the number copied */
int cuadmm_get_info_array(const cuadmm_solver* s, int which, double* out, int cap);
double cuadmm_get_total_time(const cuadmm_solver* s);
/* scalars of the current state: errRp, errRd, pobj, dobj, relgap, sig, bscale, Cscale,
 * norm_borg, norm_Corg, best_KKT, eig_not_converged (12 doubles) */
int cuadmm_get_state(const cuadmm_solver* s, double out12[12]);

/* Per-kernel-class timing collected when option "profile"=1 (HIP events on the engine
 * stream).  Classes: 0 aty_xb, 1 psd_project, 2 post_proj, 3 spmv_A, 4 copies/h2d/d2h,
 * 5 host_solve (wall, without 7), 6 allreduce, 7 GPU part of the A*A^T solve (wall, incl. transfers).  out[3*k+0]=launches, [3*k+1]=total ms, [3*k+2]=bytes
 * (algorithmic HBM bytes per launch).  */
#define CUADMM_NUM_KCLASS 8
int cuadmm_get_profile(const cuadmm_solver* s, double out[3 * CUADMM_NUM_KCLASS]);
int cuadmm_reset_profile(cuadmm_solver* s);
/* Steps of the last projection per block of this rank's shard (0 for blocks served by the eigensolver kernels); needs
 * option "psd_steps".  Returns the number of entries written (<= cap) or a negative error code. */
int cuadmm_get_psd_steps(cuadmm_solver* s, int* out, int cap);
/* Counters of the engine's execution plan (diagnostics; what bench.py reports beside its numbers):
 *   [0] launches that ran several ADMM iterations (option "batch"), [1] iterations run by them, [2] batches rolled back because
 *   the stopping test fired inside them, [3] threads of the host pool, [4] 1 if the iteration is fused into the projection
 *   kernels, [5] 1 if every block solves for its own multipliers (closed blocks), [6] 1 if the y-solve runs on the device (2: hybrid --
 *   L11 sweeps on the host, L21 products and the tail on the device; 3: on the device with dense tree tops, option "lead_tops"),
 *   [7] size of the GPU tail of the A A^T factor. */
int cuadmm_get_counters(const cuadmm_solver* s, double out8[8]);
/* The dense GPU tail of the A A^T factor on this rank: [0] its size k (0: none), [1] bytes of inv(L22) this rank read in its last
 * solve (4 K^2 for the whole triangle; 1 / world of it when the tail is sharded, option "tail_shard"), [2] the rows it applied,
 * [3] bytes of device memory the tail holds on this rank, [4] accuracy of its explicit inverse W = inv(L22) measured at init,
 * || z - L22 (W z) ||_inf for a probe vector z of entries in [0.5, 1.5] (~ unit roundoff x cond(L22); -1: not measured), [5] 1 when option
 * "tail_refine" is on: every triangular solve of the tail takes one refinement step against the factor itself (6x the bytes per solve). */
int cuadmm_get_tail_info(const cuadmm_solver* s, double out[6]);
/* The in-process group a handle leads after cuadmm_duo_init(device_num_requested = N) from one process (reference
 * src/duo_solver.cu:487-577): [0] engines in the group (1: no group), [1] exchange of its all-reduce -- 1 = device side (each
 * rank's kernel adds the N staging buffers out of its peers' memory: one shared device, or peer access over xGMI as
 * src/utils/check_gpus.cu:29-43), 0 = host-staged fallback --, [2] distinct devices, [3] all-reduces so far. */
int cuadmm_get_group_info(const cuadmm_solver* s, double out4[4]);

/* ------------------------------------------------------------------------------------ */
/* TXT problem loader: Problem::from_txt (reference src/problem.cu:11-83, src/utils/io.cu). */
/* ------------------------------------------------------------------------------------ */
typedef struct cuadmm_problem cuadmm_problem;
typedef struct {
  int vec_len, con_num, mat_num;
  int At_nnz, b_nnz, C_nnz;
  const int* At_csc_col_ptrs;
  const int* At_csc_row_ids;
  const double* At_csc_vals;
  const int* b_indices;
  const double* b_vals;
  const int* C_indices;
  const double* C_vals;
  const int* blk_vals;
} cuadmm_problem_view;

/* prefix must end in '/' exactly like the reference CLI (problem.cu:12 concatenates) */
int cuadmm_problem_from_txt(const char* prefix, cuadmm_problem** out);
int cuadmm_problem_view_get(const cuadmm_problem* p, cuadmm_problem_view* view);
void cuadmm_problem_free(cuadmm_problem* p);
/* COO_to_CSC (io.cu:187-243): sorts triplets by (col,row) in place and fills col_ptrs[col_num+1] */
int cuadmm_coo_to_csc(int* col_ptrs, int* col_ids, int* row_ids, double* vals, int nnz, int col_num);
/* read_blk (io.cu:296-329): returns number of entries; types[i] in 'a'..'z','A'..'Z' */
int cuadmm_read_blk(const char* filename, char* types, int* sizes, int cap);
/* DeviceDenseVector::to_txt format (memory.h:278-294): one "%.32f\n" per entry */
int cuadmm_write_dense_txt(const char* filename, const double* vals, int64_t n);

/* ------------------------------------------------------------------------------------ */
/* Block bookkeeping (host): analyze_blk / is_large_mat / MatrixSizes / get_maps.          */
/* ------------------------------------------------------------------------------------ */
/* is_large_mat (src/matrix_sizes.cu:14-19) */
int cuadmm_is_large_mat(int mat_size, int mat_num);
/* analyze_blk (src/utils/analyze_blk.cu:63-99): ascending unique sizes + counts; returns #sizes */
int cuadmm_analyze_blk(const int* blk, int mat_num, int* sizes_out, int* nums_out, int cap);
/* get_maps (src/utils/get_maps.cu:80-135): the reference's int32 svec->dense maps.  The
 * engine itself computes indices from (offset,n) per block; these are exported so that the
 * svec index contract can be checked bit-for-bit against the reference's test vectors. */
int cuadmm_get_maps(const int* blk, int mat_num, int vec_len, int* map_B, int* map_M1, int* map_M2);
/* get_maps_duo (get_maps.cu:21-68) */
int cuadmm_get_maps_duo(const int* blk, int mat_num, int LARGE, int SMALL, int vec_len,
                        int* map_B, int* map_M1, int* map_M2);
/* get_inverse_permutation (src/utils/inverse_permutation.cu:17-30) */
int cuadmm_inverse_permutation(const int* perm, int n, int* perm_inv);
/* contiguous block ranges per rank balanced by sum n^3; out has world+1 entries */
int cuadmm_partition_blocks(const int* blk, int mat_num, int world, int* first_block_out);
/* Rows of the dense GPU tail of the replicated y-solve (k columns, padded to K = a multiple of 64) that rank p of `world` applies
 * when the tail is sharded (option "tail_shard"; the reference splits its per-iteration heavy part over the devices,
 * src/duo_solver.cu:269-295): [out[p], out[p + 1]) in the kernels' numbering (row 0 = the longest row of the triangle), equal shares
 * of the triangle's ENTRIES, multiples of 8, out[0] = 0, out[world] = K; a range may be empty.  Host arithmetic only. */
int cuadmm_tail_shard_bounds(int k, int world, int* out);

/* ------------------------------------------------------------------------------------ */
/* Host (A A^T + eps I) factor + permuted solve: CholeskySolverCPU                        */
/* (reference include/cuadmm/cholesky_cpu.h:62-155; CHOLMOD simplicial LDL^T there).      */
/* ------------------------------------------------------------------------------------ */
typedef struct cuadmm_aat cuadmm_aat;
/* A is given in CSC (col_ptrs over the vec_len columns = the reference's At_csr arrays,
 * solver.cu:91-95).  Factors P (A A^T + eps I) P^T = L D L^T with a fill-reducing P. */
int cuadmm_aat_create(int con_num, int vec_len, const int* A_col_ptrs, const int* A_row_ids,
                      const double* A_vals, double eps, cuadmm_aat** out);
/* CHOLMOD L->Perm semantics: row i of the permuted system is original row perm[i] */
const int* cuadmm_aat_perm(const cuadmm_aat* f);
int64_t cuadmm_aat_factor_nnz(const cuadmm_aat* f);
/* column pointers (m+1) of the strict lower triangle of L, for diagnostics */
const int64_t* cuadmm_aat_factor_colptr(const cuadmm_aat* f);
/* cholmod_solve2(CHOLMOD_LDLt): NO permutation applied inside (cholesky_cpu.h:146-155);
 * caller does rhs_perm[perm_inv[i]] = rhs[i] and y[perm[i]] = sol_perm[i] (solver.cu:487,500) */
int cuadmm_aat_solve_permuted(const cuadmm_aat* f, const double* rhs_perm, double* sol_perm);
/* Dense-tail split of the solve (no reference counterpart: CHOLMOD runs both sweeps on the host).  The last k
 * columns of L are an almost dense triangle holding most of nnz(L); the engine inverts that triangle once on the GPU
 * and runs it as two GEMVs per solve, the host keeps the sparse leading columns.
 *   tail_plan            : k chosen by the cost model (0 = keep everything on the host), k <= max_k
 *   tail_dense           : trailing k x k block of L as dense row-major unit-lower matrix (leading dimension ld) and D
 *   solve_leading_forward: forward sweep over the leading m-k columns, in place; x[m-k..] then holds z2
 *   solve_leading_backward: D1^-1 and backward sweep over the leading columns, in place; x[m-k..] must hold the solved tail */
int cuadmm_aat_tail_plan(const cuadmm_aat* f, int max_k);
/* Split factorisation: like cuadmm_aat_create, but when the cost model finds a dense tail (k <= max_k) the last k
 * columns are left unfactored; the Schur complement B22 - L21 D1 L21^T (sparse, lower triangle) is kept for the
 * GPU, which scatters it into a dense matrix, factors and inverts it (tail_solve.hip).
 * max_k < 0 forces the tail size -max_k.  cuadmm_aat_solve_permuted / _tail_dense are unavailable on a split factor. */
int cuadmm_aat_create_split(int con_num, int vec_len, const int* A_col_ptrs, const int* A_row_ids,
                            const double* A_vals, double eps, int max_k, cuadmm_aat** out);
int cuadmm_aat_tail_k(const cuadmm_aat* f);
/* > 0: the cost model chose this tail for the device-side solve with DENSE TREE TOPS -- the leading elimination forest cut at this
 * height, the nodes above it solved through explicit inverses of their diagonal blocks (engine option "lead_tops") --; 0 otherwise.
 * cuadmm_aat_plan_allow_tops(0) makes the calling thread's next cuadmm_aat_create_split plan without them (the planner of round 4). */
int cuadmm_aat_tail_tops(const cuadmm_aat* f);
void cuadmm_aat_plan_allow_tops(int allow);
/* Whole (unsplit) factor for a device-side solve: the strict lower triangle of the unit factor L by columns (Lp[m+1], Li, Lx),
 * the pivots D[m], and the elimination forest as lists of columns per tree (tree t: tree_cols[tree_ptr[t] .. tree_ptr[t+1]),
 * ascending).  The sweeps of a solve never leave a tree, so a block-diagonal A A^T (one small tree per group of coupled
 * constraints) is solved by one GPU thread per tree (engine: forest solve) in the arithmetic order of the host sweeps. */
int cuadmm_aat_factor_arrays(const cuadmm_aat* f, const int64_t** Lp, const int** Li, const double** Lx, const double** D);
int cuadmm_aat_forest(cuadmm_aat* f, int* n_trees, int* max_cols, const int** tree_ptr, const int** tree_cols);
/* lower triangle with diagonal of the Schur complement by rows (CSR over the k tail rows, tail-local column indices,
 * not sorted within a row); pointers stay valid until _tail_schur_release / _free */
int cuadmm_aat_tail_schur(const cuadmm_aat* f, const int64_t** row_ptr, const int** col, const double** val);
void cuadmm_aat_tail_schur_release(cuadmm_aat* f);
int cuadmm_aat_tail_dense(const cuadmm_aat* f, int k, double* L22, int64_t ld, double* D2);
/* (the leading sweeps of ONE factor are single-caller: they share per-factor scratch; different factors are independent) */
int cuadmm_aat_solve_leading_forward(const cuadmm_aat* f, int k, double* x);
int cuadmm_aat_solve_leading_backward(const cuadmm_aat* f, int k, double* x);
/* The same sweeps restricted to L11 (rows and columns < m-k) of a SPLIT factor whose L21 the engine keeps on the GPU (hybrid solve:
 * deep leading forest, most leading nonzeros in the tail rows):
 *   forward11 : x1 <- L11^-1 x1, x[m-k..] untouched (the GPU forms z2 = x2 - L21 x1)
 *   backward11: x1 <- L11^-T (D1^-1 x1 - w), w = L21^T x2 (m-k doubles, from the GPU); x[m-k..] is not read */
int cuadmm_aat_solve_leading_forward11(const cuadmm_aat* f, int k, double* x);
int cuadmm_aat_solve_leading_backward11(const cuadmm_aat* f, int k, double* x, const double* w);
void cuadmm_aat_free(cuadmm_aat* f);

/* ------------------------------------------------------------------------------------ */
/* MATLAB front end (reference MATLAB/cuadmm_MATLAB.cu): the marshalled call behind            */
/*   [X, y, S, info] = cuadmm_MATLAB(eig_stream_num_per_gpu, max_iter, stop_tol, At, b, C, blk, X0, y0, S0, sig, ...)  */
/* Arguments are what mexFunction extracts from its mxArrays (cuadmm_MATLAB.cu:197-293): At as a sparse    */
/* vec_len x con_num matrix (size_t jc[At_cols+1], ir[nnz], pr[nnz]); b, C as sparse column vectors       */
/* (jc[2], ir, pr); blk as doubles; X0, y0, S0 dense; `optional5` = {sig_update_threshold, stage_1,        */
/* stage_2, switch_admm, sigscale} or NULL, read only under the reference's own conditions `nlhs >= 12..16`  */
/* (:297-333; nlhs <= 4, so the effective values are always 500, 50, 100, 11000 and sigscale 1.0).            */
/* The shim that calls this from a mexFunction: MATLAB/cuadmm_MATLAB_amd.cpp.                                  */
/* ------------------------------------------------------------------------------------ */
typedef struct cuadmm_mex_result cuadmm_mex_result;
int cuadmm_mex_call(int eig_stream_num_per_gpu, int max_iter, double stop_tol,
                    size_t At_rows, size_t At_cols, const size_t* At_jc, const size_t* At_ir, const double* At_pr,
                    size_t b_rows, const size_t* b_jc, const size_t* b_ir, const double* b_pr,
                    size_t C_rows, const size_t* C_jc, const size_t* C_ir, const double* C_pr,
                    size_t blk_len, const double* blk_pr,
                    size_t X0_len, const double* X0, size_t y0_len, const double* y0, size_t S0_len, const double* S0,
                    double sig, int nlhs, const double* optional5, cuadmm_mex_result** out);
/* what the MEX packs into its outputs (cuadmm_MATLAB.cu:366-424): X, y, S (unscaled) and the 10 x 2 `info` cell:
 * iter_num, the eight per-iteration arrays (which = CUADMM_INFO_*, iter_num doubles each) and total_time */
int cuadmm_mex_result_dims(const cuadmm_mex_result* r, int* vec_len, int* con_num, int* iter_num, double* total_time);
int cuadmm_mex_result_XyS(const cuadmm_mex_result* r, double* X, double* y, double* S);
int cuadmm_mex_result_info(const cuadmm_mex_result* r, int which, double* out);
void cuadmm_mex_result_free(cuadmm_mex_result* r);

/* ------------------------------------------------------------------------------------ */
/* Op-level device entry points (pointers are DEVICE pointers; `stream` is a hipStream_t   */
/* or NULL).  Each mirrors one reference kernel/wrapper so parity can be tested per op.    */
/* ------------------------------------------------------------------------------------ */
/* vector_to_matrices / matrices_to_vector (src/kernels/vec_mat_conversion.cu:11-98) */
int cuadmm_op_vector_to_matrices(const double* Xb, double* large_mat, double* small_mat,
                                 const int* map_B, const int* map_M1, const int* map_M2,
                                 int vec_len, void* stream);
int cuadmm_op_matrices_to_vector(double* Xb, const double* large_mat, const double* small_mat,
                                 const int* map_B, const int* map_M1, const int* map_M2,
                                 int vec_len, void* stream);
/* batch_eig_cusolver / single_eig_cusolver (include/cuadmm/cusolver.h:76-95,154-171):
 * `count` contiguous n x n column-major symmetric matrices, overwritten by eigenvectors
 * (column-major, column k <-> W[k]); W ascending; info[i] = 0 or 1 (iteration cap hit / orthonormalisation not converged).
 * n <= 128: one wavefront / workgroup per matrix; 129 <= n <= 8192: one matrix at a time on the whole chip (csrc/eig_large.hip:
 * tridiagonalisation, bisection, inverse iteration, Cholesky-QR; n = 2000 in 0.06 s); larger n: CUADMM_ERR_INVALID. */
int cuadmm_op_batch_eig(double* mat, double* W, int* info, int n, int count, void* stream);
/* The DGEMM of the large-block rebuild (the reference calls cublasDgemm on large_mat, src/solver.cu:630-644).
 * C = alpha * A * B + beta * E for n x n row-major SYMMETRIC A and B (device pointers, n a multiple of 64,
 * E may be null); this is the fp64 matrix-core kernel behind the large-block projection (psd_large.hip). */
int cuadmm_op_gemm_sym(int n, const double* A, const double* B, double alpha, double beta, const double* E,
                       double* C, void* stream);
/* The GPU part of the A*A^T solve on its own (tail_solve.hip): z <- L22^-T D2^-1 L22^-1 z for `nrhs` host vectors
 * of length k (contiguous), L22 dense k x k row-major unit lower triangular and D2 the pivots (host pointers). */
int cuadmm_op_tail_solve(const double* L22_host, const double* D2_host, int k, double* z2_host, int nrhs);
/* The same solve as the ranks of a sharded engine run it (TailSolve::shard_*): out_host receives `world` partial results of k doubles,
 * partial p = what rank p contributes from its rows [bounds[p], bounds[p + 1]) (cuadmm_tail_shard_bounds) -- their sum in rank order
 * is the solve; rows_out[p] = the rows rank p applied.  one_pass: bit 0 = the one-pass kernels (0: the two triangular GEMVs), bit 1 = every rank
 * keeps only its rows of inv(L22) (one object per rank, as the ranks of an engine do since round 6). */
int cuadmm_op_tail_solve_sharded(const double* L22_host, const double* D2_host, int k, const double* z_host, int world, int one_pass,
                                 double* out_host, int* rows_out);
/* Test hook: failure drill of the row-sharing tail kernel (18 432 < k <= 32 768).  out_host: 4 x k doubles -- the plain solve, the
 * solve of a right-hand side poisoned with a NaN that carries the exchange sentinel's bits, the solve with the lost-exchange
 * counter raised beforehand (NaN), the solve after the object retired to the two-pass kernels; counts[3] = {exchanges lost by the
 * poisoned solve (0), count found and cleared after the third solve (>= 1), retired flag (1)}. */
int cuadmm_op_tail_solve_drill(const double* L22_host, const double* D2_host, int k, const double* z_host, double* out_host, int* counts);
/* Same with the factorisation on the GPU too: z <- S^-1 z for a symmetric S given by its lower triangle with
 * diagonal (CSR over k rows, host pointers), dense LDL^T without pivoting (what cuadmm_aat_create_split hands over). */
int cuadmm_op_tail_factor_solve(const int64_t* row_ptr, const int* col, const double* val, int k, double* z_host, int nrhs);
/* max_dense_vector_zero (src/kernels/dense_scalar.cu:41-47,93-97) */
int cuadmm_op_max_zero(double* w, int64_t n, void* stream);
/* dense_matrix_mul_diag_batch (src/kernels/diagonal_batch.cu:11-62): out = in * diag(w) per matrix */
int cuadmm_op_mul_diag_batch(double* out, const double* in, const double* w, int n, int count, void* stream);
/* dense_matrix_mul_trans_batch (include/cuadmm/cublas.h:18-35): P = T * V^T per matrix (MFMA f64) */
int cuadmm_op_mul_trans_batch(double* P, const double* T, const double* V, int n, int count, void* stream);
/* fused PSD-cone projection over blocks laid out in svec form (solver.cu:534-647):
 * blk[mat_num] sizes in blk.txt order; Xproj may alias Xb.  eig_fail (device int, may be
 * NULL) is incremented per block whose QL iteration hit its cap. */
int cuadmm_op_psd_project(const double* Xb, double* Xproj, const int* blk_host, int mat_num, void* stream);
/* The same projection through a plan that is built once (block descriptors, size classes, workspaces on the device) and reused:
 * what the solver does every iteration; cuadmm_op_psd_project builds and drops a plan per call. */
typedef struct cuadmm_psd_plan cuadmm_psd_plan;
int cuadmm_psd_plan_create(const int* blk_host, int mat_num, int eig_rank, cuadmm_psd_plan** out);
int cuadmm_psd_plan_project(cuadmm_psd_plan* plan, const double* Xb, double* Xproj, int* steps_dev /* may be NULL */, void* stream);
void cuadmm_psd_plan_destroy(cuadmm_psd_plan* plan);
/* General form: blk[k] < 0 is an UNCONSTRAINED block of -blk[k] variables (blk.txt type 'u', reference README.md:55-64),
 * copied through; eig_rank > 0 keeps only the eig_rank largest eigenvalues of every PSD block: V diag(max(W,0) * mask) V^T with
 * the mask of get_eig_rank_mask.cu:13-37 (dense_scalar.cu:51-57) -- computed through the eigensolver kernels. */
int cuadmm_op_psd_project_ex(const double* Xb, double* Xproj, const int* blk_host, int mat_num, int eig_rank, int* steps_dev, void* stream);
/* Same, also returning how many Newton-Schulz steps the adaptive matrix-sign schedule took per block (device int array of
 * mat_num entries, 0 for blocks served by the eigensolver kernels); developer/diagnostic entry. */
int cuadmm_op_psd_project_steps(const double* Xb, double* Xproj, const int* blk_host, int mat_num, int* steps_dev, void* stream);
/* Host model of the per-block adaptive schedule of the matrix-sign projection (csrc/sign_sched.h), no device needed:
 * s[n] = |eigenvalue| / ||X||_1 on entry, the sign estimates on exit; returns the number of steps the kernels would
 * take on that spectrum and, in *max_err_out, max_i s_i(0) |1 - s_i| / 2 (projection error relative to ||X||_1).
 * lagged: 0 = the one-wavefront / one-workgroup kernels (decisions from the current iterate), 1 = the batched-GEMM path
 * (||S - S Y|| one step old), 2 = every statistic one step old (measured and rejected), 3 = as 0 with the statistics passed
 * on every step (reference for the steps on which the kernels skip them); + 8: without the mega-lift of round 5 (sign_sched.h). */
int cuadmm_sign_sched_simulate(double* s, int n, int lagged, double* max_err_out);
/* same, with the schedule's warm start across ADMM iterations: lift0 = lift steps the previous projection of the block needed
 * (0: none), *lifts_out = the hint this run leaves */
int cuadmm_sign_sched_simulate_hint(double* s, int n, int lagged, int lift0, double* max_err_out, int* lifts_out);
/* perform_permutation (src/kernels/permutation.cu:12-33): v1[perm[i]] = v2[i] */
int cuadmm_op_permute(double* v1, const double* v2, const int* perm, int n, void* stream);
/* get_normA (src/kernels/sparse_matrix_norm.cu:11-44): per CSC column norm=max(1,||col||), col/=norm */
int cuadmm_op_get_normA(const int* col_ptrs, double* vals, double* normA, int con_num, void* stream);
/* SpMV_cusparse (include/cuadmm/cusparse.h:70-83): y = alpha*A*x + beta*y, CSR int32 */
int cuadmm_op_spmv_csr(int rows, const int* row_ptrs, const int* col_ids, const double* vals,
                       const double* x, double* y, double alpha, double beta, void* stream);
/* dense_dense.cu wrappers */
int cuadmm_op_axpby2(double* v1, const double* v2, double alpha, double beta, int64_t n, void* stream);           /* v1=a*v1+b*v2 (:92-100)  */
int cuadmm_op_axpby3(double* v1, const double* v2, const double* v3, double alpha, double beta, int64_t n, void* stream); /* v1=a*v2+b*v3 (:103-111) */
/* get_norm (include/cuadmm/memory.h:238-247): 2-norm, deterministic two-stage reduction */
int cuadmm_op_norm2(const double* v, int64_t n, double* host_out, void* stream);

/* device memory helpers for bindings that do not own a device allocator */
int cuadmm_dev_malloc(void** ptr, size_t bytes);
int cuadmm_dev_free(void* ptr);
int cuadmm_memcpy_h2d(void* dst, const void* src, size_t bytes);
int cuadmm_memcpy_d2h(void* dst, const void* src, size_t bytes);
int cuadmm_dev_sync(void);

#ifdef __cplusplus
}
#endif
#endif /* CUADMM_AMD_H */
