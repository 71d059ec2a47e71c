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
  // (solver.cu:478,695,764).  lc[block]: its local rows lc.x .. lc.x + (lc.y & 0xffff) (longest: lc.y >> 16 nonzeros) and
  // their nonzeros lc.z .. lc.z + lc.w; row k: constraint lc_row[k], nonzeros lc_nzptr[k] .. lc_nzptr[k + 1]: (offset inside
  // the block's svec, value).  One LANE per nonzero forms a * v, one lane per row adds its segment in order (deterministic):
  // a block keeps its local rows only when it has at most 64 of them with at most 64 nonzeros in total (kFuseRowsMax).
  const struct LcDesc* lc; const int* lc_row; const int* lc_nzptr; const int* lc_e; const double* lc_v;
  double* outX; double* outS;                         // null: not wanted (outX only with mode 0)
  // CLOSED blocks (every constraint that touches the block is one of its local rows, at most kClosedMaxRows of them: the
  // block-diagonal configurations) run psd_sign_closed.h instead: the block also SOLVES for its own multipliers -- rhs from its
  // rows of [A X | A (S - C)] of the previous iteration, the dense unit-lower factor of its diagonal block of A A^T (cut out of
  // the host factor, same elimination order, same unfused arithmetic as forest_solve_kernel: bit-identical y; solver.cu:478-500)
  // -- scatters A^T y from those values, and in mode 0 adds its rows' share of ||Rp||^2 and b^T y (solver.cu:768-772,781) to a
  // second partial pair.  rec == null: y comes from y[] (generic fused body).
  double* y_out;                                      // y (all constraints, the factor's order)
  double* partials2;                                  // 2 doubles per fused block: sum (normA (b - A X) bscale)^2, sum b y
  double isig, bscale;
  // SEVERAL ADMM iterations in one launch (closed blocks only, mode 0): between two sigma updates nothing couples the blocks of a
  // block-diagonal problem except the four scalars of the stopping test, so a wavefront runs `iters` whole iterations on its
  // block back to back (its X, S, y and rows of [A X | A (S - C)] go through HBM / L2 exactly as between two launches) and leaves
  // one partial pair per iteration: iteration `it` writes partials + it * pstride and partials2 + it * pstride.  The engine
  // forms the scalars of all `iters` iterations afterwards (reduce_quads_batch_kernel), replays the host-side schedule on them
  // and, should the stopping test have fired inside the batch, restores its checkpoint and reruns the shorter batch -- the
  // iterates are bit-identical to one launch per iteration.  No launch ramp / tail and no host round trip per iteration.
  int iters;                                          // <= 1: one iteration
  long long pstride;                                  // doubles between the partial arrays of consecutive iterations
  // closed blocks, psd_sign_closed.h: the per-block records and the per-block copy of the block's rows of [A X | A (S - C)]
  // (16 doubles per slot: A X of row k at 16 slot + k, A (S - C) at 16 slot + 8 + k; outX / outS keep the by-row copy)
  const struct ClosedRec* rec;
  double* cl_out;
  int iter0;                                          // closed blocks: iterations run before this launch (ages the schedule hints)
  int full;                                           // every block of the launch fills its tile (n = 16 NT): set per launch by PsdPlan::project
};
struct LcDesc { int x, y, z, w; };
constexpr int kFuseRowsMax = 64;
constexpr int kClosedMaxRows = 8;

// Everything static a CLOSED block's iteration needs, in one contiguous record addressed by the block's slot alone
// (psd_sign_closed.h: one memory round trip instead of five dependent ones).  Nonzeros are ordered by local row (rows ascending
// in the factor's order); rk = row position | round << 3, where `round` counts the earlier nonzeros on the same svec slot:
// round k is applied after round k - 1, which reproduces the summation order of the CSR gather.
struct alignas(64) ClosedRec {
  int nk, nnz, nrounds, maxlen;               // local rows, their nonzeros, largest multiplicity of an svec slot, longest row
  unsigned char nzp[16];                      // nonzeros of row k: nzp[k] .. nzp[k + 1]
  int rows[kClosedMaxRows];                   // constraint index (the factor's order) of local row k
  unsigned nzt[kFuseRowsMax];                 // the nonzero's svec slot as the entry of the block's tile table (SwcTab<NT>: byte
                                              // offsets of the element and of its Rd1 slot in the LDS tile; psd_closed_tab_entry)
  unsigned char rk[kFuseRowsMax];
  double v[kFuseRowsMax];
  double L[kClosedMaxRows * kClosedMaxRows];  // unit lower factor of the block's diagonal block of A A^T, row-major
  double D[kClosedMaxRows], b[kClosedMaxRows], normA[kClosedMaxRows];
};

// the record's header, packed into PsdDesc::pad[0] (PsdPlan::set_desc_aux): the kernels know it with the descriptor
constexpr int closed_hdr_pack(int nk, int nnz, int nrounds, int maxlen) { return nk | (nnz << 4) | (nrounds << 11) | (maxlen << 18); }
__host__ __device__ constexpr int closed_hdr_nk(int h) { return h & 15; }
__host__ __device__ constexpr int closed_hdr_nnz(int h) { return (h >> 4) & 127; }
__host__ __device__ constexpr int closed_hdr_nrounds(int h) { return (h >> 11) & 127; }
__host__ __device__ constexpr int closed_hdr_maxlen(int h) { return (h >> 18) & 127; }

}  // namespace cuadmm
