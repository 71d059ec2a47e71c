// Device-side sweeps over the sparse LEADING columns of a split A A^T factor (the dense tail is tail_solve.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace cuadmm {

struct TailSolve;

// With the leading sweeps on the device the whole y-solve of a split factor runs without a PCIe hop:
//   forward  (per tree of the leading elimination forest, level by level, row gather):  z1 = L11^-1 rhs1
//   tail rhs (SpMV over the tail rows of the leading columns):                          z2 = rhs2 - L21 z1  -> TailSolve::vin
//   tail     (tail_solve.hip, two triangular GEMVs):                                    x2 = L22^-T D2^-1 L22^-1 z2
//   backward (per tree, levels in reverse, column gather):                              x1 = L11^-T (D1^-1 z1 - L21^T x2 ...)
// rhs = -A(S-C) + (b - A X) / sigma is formed on the fly from vectors resident in HBM (solver.cu:478-482).
struct LeadSolve {
  int m = 0, n1 = 0, k = 0, ntrees = 0, max_levels = 0, max_nodes = 0;
  size_t lds_bytes = 0;
  // tail rows of the leading columns (k rows, columns < n1), CSR: z2 = rhs2 - L21 z1
  long long* rp21 = nullptr; int* ci21 = nullptr; double* v21 = nullptr;
  // sweep streams in processing order (slot idx = position in nodes_*), LOCAL indices inside the tree:
  long long* fptr = nullptr; int* fci = nullptr; double* fv_ = nullptr;     // forward: row of L11 of the node in the slot
  long long* bptr = nullptr; int* bci = nullptr; double* bv_ = nullptr;     // backward: leading rows of its column
  long long* tptr = nullptr; int* tri = nullptr; double* tv_ = nullptr;     // tail rows of every leading column (w = L21^T x2)
  int* long_cols_d = nullptr;   // leading columns with more than 128 tail rows: a wavefront each in w = L21^T x2 (lead_l21t_long_kernel)
  int n_long = 0;
  double* D1 = nullptr;
  double* wvec = nullptr;
  // trees: nodes ordered by level inside each tree, per sweep direction
  int* nodes_f = nullptr; int* nodes_b = nullptr;
  int* lvl_ptr_f = nullptr; int* lvl_ptr_b = nullptr;      // offsets into lvl_off_* per tree (ntrees + 1)
  int* lvl_off_f = nullptr; int* lvl_off_b = nullptr;      // level boundaries in nodes_* (levels + 1 entries per tree)
  int* lvl_g_f = nullptr; int* lvl_g_b = nullptr;          // lanes per row of each level (power of two, from its mean row length)
  // trees by the LDS they need with their stream resident: <= 16 KB | up to one workgroup's LDS | streaming kernels
  void *desc_small_f = nullptr, *desc_small_b = nullptr, *desc_big_f = nullptr, *desc_big_b = nullptr;   // LeadTreeDesc per tree and sweep
  int* trees_stream = nullptr;
  int n_small = 0, n_big = 0, n_stream = 0;
  size_t lds_small = 0, lds_big = 0;
  // MICRO trees: one or two nodes (bqp-r1-40-1 at a 1 024-column tail: 102 000 of its 103 016 trees).  A wavefront per tree is the wrong shape for
  // them; when there are thousands, one THREAD takes a tree (lead_micro_kernel) -- a row of such a tree has at most one entry, so the sums are
  // the wavefront kernels' bit for bit
  int* micro_first = nullptr;   // first slot of every micro tree (its size: 1 or 2, in micro_cnt)
  int* micro_cnt = nullptr;
  int n_micro = 0;
  hipStream_t aux = nullptr;    // the big trees' launches run beside the small trees' (fork / join events on the caller's stream)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool debug = false;           // option lead_debug: forest statistics on stderr at build
  int small_kb = 0;             // option lead_small_kb: trees that need at most this much LDS share a workgroup in fours (one wavefront each); 0: chosen at build
  double pinv_tol = 0.0;        // option pinv_tol (experiment): leading pivots below it in magnitude are treated as infinite (1 / d := 0), like the tail's
  bool stream_only = false;     // option lead_stream = 1: every tree on the streaming kernels (A/B, tests)
  bool ready = false;
  double est_us = 0;            // cost model used to decide (per solve)
  // HYBRID solve: the forest is too deep for the device (or its sweeps lose to the host's), but most leading nonzeros sit in the tail
  // rows (PlanarHand_N=10 at k = 32 768: L11 13.7 M, L21 31.9 M).  Then only L21 lives here: the host sweeps L11
  // (cuadmm_aat_solve_leading_forward11 / _backward11), the device forms z2 = rhs2 - L21 z1, solves the tail and returns w = L21^T x2.
  bool hybrid = false;
  long long nnz21 = 0;
  double* zfull = nullptr;      // device, m doubles: [z1 | rhs2] of the current solve
  double* h_w = nullptr;        // pinned, n1 doubles: w = L21^T x2
  double* h_z = nullptr;        // pinned, m doubles: staging for a caller vector that is not page-locked
  // per solve: the host saves two passes over L21 (~0.4 ns per nonzero on the pool), the device pays two m-vectors over PCIe and two SpMVs
  bool force_hybrid = false;    // option l21_device = 2 (tests, A/B)
  bool l21_pays() const {
    if (force_hybrid) return k > 0 && n1 > 0;
    const double save_us = 0.8e-3 * (double)nnz21;
    const double cost_us = 2.0 * (double)m * 8.0 / 20e3 + 2.0 * (double)nnz21 * 12.0 / 3.5e6 + 60.0;
    return k > 0 && n1 > 0 && save_us > 2.0 * cost_us;
  }
  // after build(): drop everything but L21 (the engine rejected the device-side sweeps); false when L21 on the device does not pay
  bool demote_to_hybrid();
  // x (host, m doubles): in [z1 | rhs2] (z1 = L11^-1 rhs1), out [z1 | x2]; h_w <- L21^T x2.  Synchronous on `st`.
  int apply_l21(double* x, bool x_pinned, TailSolve& tail, hipStream_t st);

  // DENSE TREE TOPS (round 5; lead_solve.hip: build_tops).  A forest that is too deep for level-by-level sweeps (PlanarHand_N=10 at a
  // 32 768-column tail: 36 big trees, 1 046 levels) is cut at height `tops_level`: the nodes at or above it -- the top T of every big
  // tree, an upward-closed set -- leave the sweeps and become a block-diagonal MIDDLE stage with one explicit dense inverse per tree,
  //     [ L_BB            ]       forward:  z_B = L_BB^-1 r_B (sweeps over the shallow rest B),  z_T = W_T (r_T - L_TB z_B),  W_T = L_TT^-1
  //     [ L_TB  L_TT      ]                 z_K = r_K - L_KB z_B - L_KT z_T,  the tail,  and back in mirror order
  //     [ L_KB  L_KT  L_KK]
  // in a symmetric permutation B | T | K of the factor's order (lower triangular again: whatever a T column reaches is in T or K).
  // Inside, everything is indexed in that order (`rid`: position -> row of the caller's vectors); `k` is then |T| + the tail.
  int tops_level = -1;          // option lead_tops: -1 = when the forest is too deep for the sweeps, 0 = never, L > 0 = always, cut at height L
  bool tops = false;
  int nT = 0, k_tail = 0;
  int* rid = nullptr;           // m: position in B | T | K order -> original index
  double* xp = nullptr;         // m: the sweeps' vector in B | T | K order
  double *zext = nullptr, *xext = nullptr, *zT = nullptr, *uT = nullptr, *DT = nullptr;   // nT + tail | same | nT | nT | nT
  double *Wf = nullptr, *Wb = nullptr;                   // rows of W_T (lower, packed) and of W_T^T (upper, packed), block after block
  long long *wf_off = nullptr, *wb_off = nullptr;        // per T row: where its packed row starts
  int *wf_beg = nullptr, *wb_len = nullptr;              // per T row: first column of its block (row i of W spans wf_beg[i] .. i), length of its row of W^T
  long long *kt_rp = nullptr, *tk_cp = nullptr;          // L_KT by tail rows (CSR) and by T columns (CSC), T-local / tail-local indices
  int *kt_ci = nullptr, *tk_ri = nullptr;
  double *kt_v = nullptr, *tk_v = nullptr;
  // optional: one step of iterative refinement per direction with the SPARSE L_TT (z = W r; z += W (r - L z)).  Built to test whether the
  // explicit inverses are what moves PushBox_N=50's primal objective by 9e-8 from the oracle's at a tail of 8 448 columns: they are not --
  // with and without the step 8.0e-8 / 9.2e-8, and 9.0e-8 with the WHOLE leading part swept on the host at that tail (profiles/
  // r05_tops_deviation.txt).  The inverses are benign (max |W| = 2.3); the deviation follows the tail's boundary among the 9 301 pivots
  // at the regularisation (1e-15: dependent constraints; 5e-9 at 8 192 columns, 4e-8 at 8 704).  Off by default.
  long long *tt_rp = nullptr, *tt_cp = nullptr;          // L_TT strictly lower by rows (CSR) and by columns (CSC), T-local
  int *tt_ci = nullptr, *tt_ri = nullptr;
  double *tt_v = nullptr, *tt_cv = nullptr;
  bool tops_refine = false;     // option lead_tops_refine
  long long tops_bytes = 0;
  int tops_blocks = 0, tops_max = 0;
  int solve_tops(const double* ax, const double* asmc, const double* b, double isig, double* y, TailSolve& tail, hipStream_t st) const;

  // Lp / Li / Lx / D: the split factor (cuadmm_aat_factor_arrays), k = its tail size
  int build(int m, int k, const int64_t* Lp, const int* Li, const double* Lx, const double* D, bool allow_hybrid = false);
  // y <- (L D L^T)^-1 (-asmc + (b - ax) * isig): everything on `st`, nothing synchronises
  int solve(const double* ax, const double* asmc, const double* b, double isig, double* y, TailSolve& tail, hipStream_t st) const;
  void release();
 private:
  int build_core(int m, int k, const int64_t* Lp, const int* Li, const double* Lx, const double* D, bool allow_hybrid);
  int build_tops(int m, int k, const int64_t* Lp, const int* Li, const double* Lx, const double* D, int level);
 public:
  ~LeadSolve() { release(); }
};

}  // namespace cuadmm
