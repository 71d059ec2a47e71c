// Dense tail of the A A^T solve on the GPU (tail_solve.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace cuadmm {

struct TailSolve {
  int k = 0, K = 0;            // tail size, padded to a multiple of 64
  double *W = nullptr, *Wt = nullptr, *dinv = nullptr, *vin = nullptr, *vmid = nullptr;   // device
  double* h_vec = nullptr;     // pinned staging vector
  double* xpart = nullptr;     // one-pass variant: n_wg partial result vectors
  int n_wg = 0;
  int* d_fail = nullptr;       // raised by a workgroup of that kernel that waited seconds for its partners in vain
  int fail_count(hipStream_t st);
  int take_failure(hipStream_t st);      // fail_count, then: counter cleared, exchange slots reset, group kernel retired
  bool group_retired = false;  // a row exchange was lost once: apply() uses the two triangular GEMVs from then on
  unsigned long long* part = nullptr;   // 18 432 < K <= 32 768: per row and member, the exchanged parts of u = W z (ts_onepass_group_kernel)
  bool attr_set = false;       // the one-pass kernel's LDS attribute has been raised
  // option tail_refine (experiment, NOTEBOOK.md "Round 6"): one step of iterative refinement of each TRIANGULAR solve against
  // the factor itself, u <- u + W (z - L u) and x <- x + W^T (v - L^T x) -- W = inv(L) is only as accurate as cond(L) allows, and L carries columns
  // of size 1 / sqrt(pivot) where the Schur complement is nearly singular; L and L^T are then kept beside W and W^T (two more K x K matrices)
  bool refine = false;
  double inv_resid = -1.0;     // || z - L (W z) ||_inf for a probe z at build: the accuracy of the explicit inverse (-1: not measured, e.g. build() from a host factor)
  double *Lm = nullptr, *Lt = nullptr, *t1 = nullptr, *t2 = nullptr;
  int apply_refined(double* v, hipStream_t st);
  // option tail_pivot (round 6, default on): the dense LDL^T of the Schur complement with diagonal pivoting (tail_solve.hip, ts_ldlt_factor_pivoted):
  // P S P^T = L D L^T, perm_d[a] = the tail-local index at position a.  The one-pass kernels gather z through it and scatter x back; the other
  // paths go through the two scratch vectors pv1 / pv2.  Null after build() from a host factor.
  bool pivot = true;
  int* perm_d = nullptr;
  int* pinv_d = nullptr;       // its inverse: a producer that writes z[pinv_d[i]] (z_scatter()) saves the one-pass kernels their gather -- 640 scattered 8-byte
                               // requests per wavefront before the first dot product -- for the same number of stores
  bool vin_pivot = false;      // set by such a producer for the NEXT solve_device only
  bool linear_z_ok() const;
  const int* z_scatter() const;
  double *pv1 = nullptr, *pv2 = nullptr;
  bool dd_dot = false;         // option tail_dd (experiment): u = W z accumulated in double-double (K <= 10 240 and 14 336 < K <= 16 384 only)
  bool fat = false;            // option tail_fat (measured, off): K <= 10 240 on 512-thread workgroups of up to 256 VGPRs (two rows per group, two groups in
                               // flight): 92 us against 81 at K = 9 216, 48 against 47 at 7 168 -- NOTEBOOK.md "Round 6"
  int depth = 1;               // option tail_depth: row groups in flight beyond the current one in the one-pass kernel (tail_solve.hip: ts_onepass_kernel)
  bool group_pf = false;       // option tail_group_pf (measured, off): the row-sharing kernel (24 576 < K <= 32 768) with two rows per exchange and the next group in flight across it -- PlanarHand N = 10 4.25 -> 4.47 ms per iteration
  bool zreg = true;            // option tail_zreg: the one-pass kernel keeps a thread's entries of z in registers where they fit (0: in LDS, rounds 3 - 6)
  int rows_per_group = 0;      // option tail_rb: rows that share one barrier in the one-pass kernel (0: two up to 8 192 columns, else one)
  int order = 2;               // option tail_order: 1 = a workgroup walks its rows alternately from the long and the short end, 2 = and odd workgroups start at the short end (0: longest first, rounds 3 - 6)
  bool prefetch = true;        // option tail_prefetch: the one-pass kernel keeps the next rows in flight across its barrier (0: rounds 3 - 5)
  bool one_pass = true;        // option tail_one_pass: x = W^T D^-1 W z in one pass over W (0: two triangular GEMVs)
  double pinv_tol = 0.0;       // option tail_pinv_tol (experiment, DESIGN.md section 4 "Round 5: dense tree tops"): pivots of the tail below it in
                               // magnitude are treated as zero (1 / d := 0); confirms where the pobj deviations of the moment relaxations come from, fixes nothing
  int apply(hipStream_t st);   // vin <- W^T diag(dinv) W vin
  // Rows of W split over the ranks of a sharded engine (reference's device split: src/duo_solver.cu:269-295): rank p applies the rows
  // of its share of the triangle (equal numbers of entries) and the K partial results are summed by reduce_fn -- the engine's
  // all-reduce, on the stream --, so a rank reads 1 / world of the 4 K^2 bytes per solve.  world = 1: the whole triangle, no reduction.
  int shard_rank = 0, shard_world = 1;
  int (*reduce_fn)(void* user, double* buf, size_t count, hipStream_t st) = nullptr;
  void* reduce_user = nullptr;
  int shard_rows = 0;          // rows this rank applied in the last solve
  double shard_bytes = 0;      // bytes of W it read for them
  double resident_bytes = 0;   // device memory held by this object
  int apply_rows(hipStream_t st, int r_begin, int r_end);
  // Round 6: a rank of a sharded engine KEEPS only the rows of W it applies (the reference splits its buffers over the devices too,
  // src/duo_solver.cu:269-295): after build, keep_shard(rank, world) copies rows [K - r_end, K - r_begin) of W into a compact matrix of their own
  // width (the triangle's rows end at the diagonal) and frees W and W^T -- 1 / world of the triangle's bytes per rank instead of two K x K squares.
  // From then on the object applies that range only (shard_rank / shard_world are fixed); the two-GEMV fallback's second pass becomes a
  // column accumulation over the same rows.  Not with option tail_refine (every rank then applies the whole tail).
  int keep_shard(int rank, int world, hipStream_t st);
  bool compact = false;
  double* Wc = nullptr;        // the kept rows; W then points to Wc - i_lo * ldc (row i of the triangle at W + i * ldc), never dereferenced outside them
  long long ldc = 0;
  int i_lo = 0, i_hi = -1;
  double build_s = 0, factor_s = 0;
  int build(const double* L22, const double* D2, int k, hipStream_t st);                                        // host factor
  int build_from_schur(const long long* row_ptr, const int* col, const double* val, int k, hipStream_t st);    // GPU factor
  int alloc(int k);
  int invert(const double* dL, hipStream_t st);
  int solve(double* z2, hipStream_t st);
  int solve_device(hipStream_t st);      // in place on the device vector `vin` (k entries, zero padding kept); asynchronous
  void release();
  ~TailSolve() { release(); }
};

// dense building blocks (row-major K x K device matrices, K a multiple of 64; asynchronous on `st`), shared with eig_large.hip
int ts_gemm(int M, int N, int Kd, double alpha, const double* A, long long lda, long long sA, const double* B, long long ldb,
            long long sB, double* C, long long ldc, long long sC, int batch, hipStream_t st);      // C = alpha A B (batched by strides)
int ts_transpose(const double* src, double* dst, int K, hipStream_t st);
int ts_ldlt_factor(double* dS, int K, double* dd, double* Yp, int* dflag, hipStream_t st);         // in place, lower triangle
int ts_unit_lower_inverse(const double* dL, double* W, double* dT, int K, hipStream_t st);         // W = inv(L), dT: scratch

}  // namespace cuadmm
