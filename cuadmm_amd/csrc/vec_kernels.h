// Launchers of the HBM-bound iteration kernels (definitions in vec_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace cuadmm {

// svec rows of At with more than `cap` entries (slots shared by thousands of constraints): one workgroup each
struct AtyLongRows {
  int cap = 128;
  int nlong = 0;
  int max_short = 0;     // longest row below the cap (chooses the lanes-per-row variant of aty_xb)
  int* rows = nullptr;   // device
  int build(long long L, const int* rp_host);
  void release();
  ~AtyLongRows() { release(); }
};

// Rd1 = At*y - C ; optionally Xb = X + sig*Rd1.   At in CSR over the L svec rows.
int launch_aty_xb(bool write_xb, long long L, const int* rp, const int* ci, const double* av, const double* y,
                  const double* C, const double* X, double sig, double* Rd1, double* Xb, hipStream_t st,
                  const AtyLongRows* long_rows = nullptr);

// mode 0: S, Rd, X update, sums | mode 1: S only | mode 2: Rd, X update, sums.
// sums_out[0] = sum Rd^2, sums_out[1] = sum C.*X (device pointer); partials: 2*post_grid(L) doubles.
int post_grid(long long L);
int launch_post(int mode, long long L, const double* Xproj, const double* Rd1, const double* C, double* X, double* S,
                double inv_sig, double tau_sig, double* partials, double* sums_out, hipStream_t st);

// sGS second half in one pass: Rd1 = At*y - C (not stored), Rd = Rd1 + S, X += tau_sig*Rd, sums (no long rows of At)
int launch_aty_post2(long long L, const int* rp, const int* ci, const double* av, const double* y, const double* C, const double* S, double* X,
                     double tau_sig, double* partials, double* sums_out, hipStream_t st);

// Fused iteration (psd_fuse.h): the same two steps restricted to a list of svec rows (those outside the fused blocks); the
// mode-0 reduction also folds in the per-block partial pairs the projection kernels wrote to partials[0, 2 nfused).
// partials: 2 * (nfused + post_grid(nidx)) doubles.
int launch_aty_xb_idx(long long nidx, const int* idx, const int* rp, const int* ci, const double* av, const double* y, const double* C,
                      const double* X, double sig, double* Rd1, double* Xb, hipStream_t st);
int launch_post_rest(int mode, long long nidx, const int* idx, int nfused, const double* Xproj, const double* Rd1, const double* C, double* X,
                     double* S, double inv_sig, double tau_sig, double* partials, double* sums_out, hipStream_t st, int* nparts_out = nullptr);
// all four scalars of the stopping test from the per-block partial pairs of a fused iteration with closed blocks
int reduce_quads_segments(int n1, int n2);
int launch_reduce_quads(const double* p1, int n1, const double* p2, int n2, double* out4, double* sums_out, double* seg_scratch, hipStream_t st);
// per-block copy of the closed blocks' rows of [A X | A (S - C)] from the by-row vectors (psd_sign_closed.h)
struct ClosedRec;
int launch_closed_gather_out(const ClosedRec* rec, int nslots, const double* ax, const double* as, double* cl_out, hipStream_t st);
// several iterations per launch: out4[4 k ..] from the partial arrays of iteration k (p1 + k stride, p2 + k stride; n pairs each)
int launch_reduce_quads_batch(const double* p1, const double* p2, int n, long long stride, int iters, double* out4, hipStream_t st);

// Rows of A with more than `cap` nonzeros (a trace / all-ones constraint): their tail is summed in segments by extra
// workgroups and added in segment order (reproducible), so one row cannot serialise the SpMV.
struct SpmvLongRows {
  int cap = 256, seg_len = 4096;   // cap is set by build(): max(256, 8 x average row length)
  int nlong = 0, nseg = 0;
  int *long_row = nullptr, *long_seg0 = nullptr, *seg_begin = nullptr, *seg_end = nullptr;   // device
  double* partial = nullptr;                                                                 // device, 2 per segment
  int build(int rows, const int* rp_host);
  void release();
  ~SpmvLongRows() { release(); }
};

// outX = A*X, outS = A*(S-C) over the rows of A (either output may be null)
int launch_spmv_rows(int rows, double avg_nnz, const int* rp, const int* ci, const double* av, const double* X,
                     const double* S, const double* C, double* outX, double* outS, hipStream_t st,
                     const SpmvLongRows* long_rows = nullptr, const int* rowmap = nullptr);   // rowmap: compact row -> output slot

int launch_scale(double* v, long long n, double s, hipStream_t st);
int launch_copy(double* dst, const double* src, long long n, hipStream_t st);   // device to device, 16-byte aligned pointers
struct CopyJobs { void* dst[6]; const void* src[6]; long long nbytes[6]; int count; };   // lengths: multiples of 4 bytes
int launch_copy_multi(const CopyJobs& jobs, hipStream_t st);                    // all of them in one launch
// y = (L D L^T)^-1 (-A(S-C) + (b - A X) / sigma) on the device, one thread per tree of the elimination forest
int launch_forest_solve(int ntrees, const int* tree_ptr, const int* tree_cols, const long long* Lp, const int* Li, const double* Lx,
                        const double* D, const double* ax, const double* asmc, const double* b, double isig, double* x, hipStream_t st);
// owned-constraints sharding: [||Rp org||^2, b.y, sums[0], sums[1]] from A*X on the device (one workgroup, deterministic)
int launch_rp_stats(int m, const double* ax, const double* b, const double* normA, const double* y, double bscale,
                    const double* sums, double* partials /* 128 doubles */, double* out4, hipStream_t st);

}  // namespace cuadmm
