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
};

}  // namespace cuadmm
