// C entry behind the MATLAB MEX function: takes exactly what the reference's mexFunction marshals out of its mxArrays
// (MATLAB/cuadmm_MATLAB.cu:197-333 of the reference) and returns what it packs into [X, y, S, info] (:342-424).
//
//   [X, y, S, info] = cuadmm_MATLAB(eig_stream_num_per_gpu, max_iter, stop_tol, At, b, C, blk, X0, y0, S0, sig, ...)
//
// MATLAB hands sparse matrices over as size_t jc / ir arrays (mxGetJc / mxGetIr); the reference casts them to int32 on
// the GPU (long_int_to_int, cuadmm_MATLAB.cu:43-88) -- here on the host, with a range check.  `blk` arrives as doubles
// (:245-256).  Effective defaults, replicated on purpose: the reference reads its five optional arguments only when
// `nlhs >= 12 ... 16` (:297-333) -- nlhs is the number of OUTPUTS, at most 4 -- so a MATLAB call always runs with
// sig_update_threshold = 500, stage_1 = 50, stage_2 = 100, switch_admm = 11000 and sigscale = 1.0 (NOT the 1.05 of
// SDPSolver::solve), whatever the caller passes; X0, y0, S0 are always given to init (warm start) and solve runs with
// if_first = true (:345-363).  The shim source that calls this from a real mexFunction is MATLAB/cuadmm_MATLAB_amd.cpp.
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "common.h"
#include "cuadmm_amd.h"

using namespace cuadmm;

struct cuadmm_mex_result {
  int vec_len = 0, con_num = 0, iter_num = 0;
  double total_time = 0;
  std::vector<double> X, y, S;
  std::vector<double> info[8];
};

namespace {

bool to_i32(const size_t* src, size_t n, std::vector<int>& dst, const char* what) {
  dst.resize(n);
  for (size_t i = 0; i < n; ++i) {
    if (src[i] > (size_t)std::numeric_limits<int>::max()) { set_error("mex_call: %s[%zu] = %zu does not fit int32", what, i, src[i]); return false; }
    dst[i] = (int)src[i];
  }
  return true;
}

}  // namespace

extern "C" {

int cuadmm_mex_call(int eig_stream_num_per_gpu, int max_iter, double stop_tol,
                    size_t At_rows, size_t At_cols, const size_t* At_jc, const size_t* At_ir, const double* At_pr,
                    size_t b_rows, const size_t* b_jc, const size_t* b_ir, const double* b_pr,
                    size_t C_rows, const size_t* C_jc, const size_t* C_ir, const double* C_pr,
                    size_t blk_len, const double* blk_pr,
                    size_t X0_len, const double* X0, size_t y0_len, const double* y0, size_t S0_len, const double* S0,
                    double sig, int nlhs, const double* optional5, cuadmm_mex_result** out) {
  if (!out) { set_error("mex_call: null result pointer"); return CUADMM_ERR_INVALID; }
  *out = nullptr;
  if (!At_jc || !b_jc || !C_jc || !blk_pr || !X0 || !y0 || !S0 || (!At_ir && At_jc[At_cols]) || (!At_pr && At_jc[At_cols])) {
    set_error("mex_call: null argument");
    return CUADMM_ERR_INVALID;
  }
  if (At_rows > (size_t)std::numeric_limits<int>::max() || At_cols > (size_t)std::numeric_limits<int>::max()) {
    set_error("mex_call: At is %zu x %zu: dimensions do not fit int32", At_rows, At_cols);
    return CUADMM_ERR_INVALID;
  }
  const int vec_len = (int)At_rows, con_num = (int)At_cols;                   // cuadmm_MATLAB.cu:218-221
  // the reference's asserts (:232,:244,:258,:268,:278,:288; compiled out in its release build) as error returns
  if (b_rows != At_cols) { set_error("mex_call: b has %zu rows, At has %zu columns", b_rows, At_cols); return CUADMM_ERR_INVALID; }
  if (C_rows != At_rows) { set_error("mex_call: C has %zu rows, At has %zu rows", C_rows, At_rows); return CUADMM_ERR_INVALID; }
  if (X0_len != At_rows || S0_len != At_rows || y0_len != At_cols) {
    set_error("mex_call: X0 / y0 / S0 have %zu / %zu / %zu entries, expected %zu / %zu / %zu", X0_len, y0_len, S0_len, At_rows, At_cols, At_rows);
    return CUADMM_ERR_INVALID;
  }
  std::vector<int> blk(blk_len);
  long long from_blk = 0;
  for (size_t i = 0; i < blk_len; ++i) {
    blk[i] = (int)blk_pr[i];                                                   // :252-255
    from_blk += blk_svec_len(blk[i]);
  }
  if (from_blk != (long long)vec_len) { set_error("mex_call: blk describes %lld svec entries, At has %d rows", from_blk, vec_len); return CUADMM_ERR_INVALID; }
  std::vector<int> At_cp, At_ri, b_idx, C_idx;
  const size_t At_nnz = At_jc[At_cols], b_nnz = b_jc[1], C_nnz = C_jc[1];
  if (!to_i32(At_jc, At_cols + 1, At_cp, "At.jc") || !to_i32(At_ir, At_nnz, At_ri, "At.ir") || !to_i32(b_ir, b_nnz, b_idx, "b.ir") ||
      !to_i32(C_ir, C_nnz, C_idx, "C.ir"))
    return CUADMM_ERR_INVALID;

  // optional arguments: read only under the reference's own (never true) conditions
  int sig_update_threshold = 500, sig_update_stage_1 = 50, sig_update_stage_2 = 100, switch_admm = (int)1.1e4;
  double sigscale = 1.0;
  if (optional5) {
    if (nlhs >= 12) sig_update_threshold = (int)optional5[0];
    if (nlhs >= 13) sig_update_stage_1 = (int)optional5[1];
    if (nlhs >= 14) sig_update_stage_2 = (int)optional5[2];
    if (nlhs >= 15) switch_admm = (int)optional5[3];
    if (nlhs >= 16) sigscale = optional5[4];
  }

  cuadmm_solver* s = nullptr;
  int rc = cuadmm_create(&s);
  if (rc) return rc;
  const int cpu_eig_thread_num = -1;                                           // "inactive parameter" (:343)
  rc = cuadmm_init(s, eig_stream_num_per_gpu, cpu_eig_thread_num, vec_len, con_num, At_cp.data(), At_ri.data(), At_pr, (int)At_nnz,
                   b_idx.data(), b_pr, (int)b_nnz, C_idx.data(), C_pr, (int)C_nnz, blk.data(), (int)blk_len, X0, y0, S0, sig);
  if (!rc) rc = cuadmm_solve(s, max_iter, stop_tol, sig_update_threshold, sig_update_stage_1, sig_update_stage_2, switch_admm, sigscale, 1);
  if (rc) { cuadmm_destroy(s); return rc; }
  cuadmm_mex_result* r = new cuadmm_mex_result();
  r->vec_len = vec_len; r->con_num = con_num;
  r->X.resize((size_t)vec_len); r->y.resize((size_t)con_num); r->S.resize((size_t)vec_len);
  rc = cuadmm_get_X(s, r->X.data());
  if (!rc) rc = cuadmm_get_y(s, r->y.data());
  if (!rc) rc = cuadmm_get_S(s, r->S.data());
  r->iter_num = cuadmm_get_info_iter_num(s);
  r->total_time = cuadmm_get_total_time(s);
  for (int w = 0; w < 8 && !rc; ++w) {
    r->info[w].resize((size_t)r->iter_num);
    const int n = cuadmm_get_info_array(s, w, r->info[w].data(), r->iter_num);
    if (n < 0) rc = n;
  }
  cuadmm_destroy(s);
  if (rc) { delete r; return rc; }
  *out = r;
  return CUADMM_OK;
}

int cuadmm_mex_result_dims(const cuadmm_mex_result* r, int* vec_len, int* con_num, int* iter_num, double* total_time) {
  if (!r) { set_error("mex_result: null"); return CUADMM_ERR_INVALID; }
  if (vec_len) *vec_len = r->vec_len;
  if (con_num) *con_num = r->con_num;
  if (iter_num) *iter_num = r->iter_num;
  if (total_time) *total_time = r->total_time;
  return CUADMM_OK;
}
int cuadmm_mex_result_XyS(const cuadmm_mex_result* r, double* X, double* y, double* S) {
  if (!r) { set_error("mex_result: null"); return CUADMM_ERR_INVALID; }
  if (X) std::memcpy(X, r->X.data(), sizeof(double) * r->X.size());
  if (y) std::memcpy(y, r->y.data(), sizeof(double) * r->y.size());
  if (S) std::memcpy(S, r->S.data(), sizeof(double) * r->S.size());
  return CUADMM_OK;
}
int cuadmm_mex_result_info(const cuadmm_mex_result* r, int which, double* out) {
  if (!r || !out || which < 0 || which >= 8) { set_error("mex_result_info: bad argument"); return CUADMM_ERR_INVALID; }
  std::memcpy(out, r->info[which].data(), sizeof(double) * r->info[which].size());
  return (int)r->info[which].size();
}
void cuadmm_mex_result_free(cuadmm_mex_result* r) { delete r; }

}  // extern "C"
