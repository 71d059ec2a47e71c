// Arguments of the FUSED projection kernels (psd_sign_wave.h), shared with the engine and the planner.
#pragma once

namespace cuadmm {

// The svec-length vector work of one ADMM iteration, FUSED around the projection of a block (engine.hip, PsdPlan::project):
//   prologue  Rd1 = A^T y - C,  Xb = X + sigma Rd1            (aty_xb_kernel; src/solver.cu:514-527) -- Xb never reaches HBM
//   epilogue  S = (P(Xb) - X) / sigma - Rd1  [mode 1 stops here]
//             Rd = Rd1 + S,  X += tau sigma Rd,  sum Rd^2,  <C, X>   (post_kernel<0>; solver.cu:652-656,746-758,774-776)
// Same expressions, element by element, as the stand-alone kernels; only the order of the two sums differs (one partial
// pair per block, summed in block-slot order: still run-to-run deterministic).  The projection kernels are bound by the
// matrix cores and leave HBM idle, so this traffic (68 B per svec element instead of 100) hides behind other blocks' MFMAs.
struct SignFuse {
  const int* rp; const int* ci; const double* av;   // A^T in CSR over the svec rows (columns: constraints in the factor's order)
  const double* y; const double* C;
  double* X; double* Rd1; double* S;
  double* partials;                                   // 2 doubles per fused block (mode 0)
  double sig, inv_sig, tau_sig;
  int mode;                                           // 0: S, X update, sums;  1: S only
  // Constraint rows whose nonzeros all lie in ONE fused block ("local" rows: every row of a block-diagonal problem) are
  // evaluated by that block's kernel from the values it holds, instead of gathering them back from HBM in spmv_rows_kernel
  // (3 random 8-byte gathers per nonzero: 128 MB moved for 15 MB at C2): outX[row] = sum a X_new, outS[row] = sum a (S - C)
  // (solver.cu:478,695,764).  lc_ptr[slot] .. lc_ptr[slot + 1]: the local rows of the block in partial-sum slot `slot`;
  // row k: constraint lc_row[k], nonzeros lc_nzptr[k] .. lc_nzptr[k + 1]: (offset inside the block's svec, value).  One LANE
  // per nonzero forms a * v into the free part of the tile, one lane per row adds its segment in order (deterministic); the
  // host only marks rows local when the block's nonzeros fit there (fuse_rows_capacity).
  const int* lc_ptr; const int* lc_row; const int* lc_nzptr; const int* lc_e; const double* lc_v;
  double* outX; double* outS;                         // null: not wanted (outX only with mode 0)
};

// doubles of LDS left for the products next to the packed svec of a block of size n (tile of the kernel that serves it)
inline int fuse_rows_capacity(int n) {
  const int np = n <= 16 ? 16 : (n <= 32 ? 32 : (n <= 48 ? 48 : 64));
  return np * (np + 1) - n * (n + 1) / 2;
}

}  // namespace cuadmm
