// MEX front end of the MI355X engine: the drop-in for the reference's MATLAB/cuadmm_MATLAB.cu (same call, same outputs):
//
//   [X, y, S, info] = cuadmm_MATLAB(eig_stream_num_per_gpu, max_iter, stop_tol, At, b, C, blk, X0, y0, S0, sig, ...)
//
// All work happens behind the C ABI (include/cuadmm_amd.h, cuadmm_mex_call); this file only moves mxArrays in and out,
// in the order of the reference's INPUT_ID / OUTPUT_ID / OUTPUT_INFO_RID tables (cuadmm_MATLAB.cu:98-183).
// Build (needs MATLAB's mex.h; this image has none, so the file is compiled only where it exists):
//   mex -I<repo>/include MATLAB/cuadmm_MATLAB_amd.cpp -L<repo>/cuadmm_amd/lib -lcuadmm_amd -output cuadmm_MATLAB
// tests/test_gpu_mex.py drives cuadmm_mex_call through ctypes with the same marshalled arguments.
#include <cstring>
#include <vector>

#include "mex.h"
#include "matrix.h"

#include "cuadmm_amd.h"

namespace {
const char* kInfoNames[10] = {"iter_num", "pobj_arr", "dobj_arr", "errRp_arr", "errRd_arr", "relgap_arr",
                              "sig_arr", "bscale_arr", "Cscale_arr", "total_time"};   // cuadmm_MATLAB.cu:157-183
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  if (nrhs < 11) mexErrMsgIdAndTxt("cuadmm:nrhs", "cuadmm_MATLAB needs 11 inputs: eig_stream_num_per_gpu, max_iter, stop_tol, At, b, C, blk, X0, y0, S0, sig");
  const int eig_streams = (int)mxGetScalar(prhs[0]);
  const int max_iter = (int)mxGetScalar(prhs[1]);
  const double stop_tol = mxGetScalar(prhs[2]);
  const mxArray *At = prhs[3], *b = prhs[4], *C = prhs[5], *blk = prhs[6], *X0 = prhs[7], *y0 = prhs[8], *S0 = prhs[9];
  if (!mxIsSparse(At) || !mxIsSparse(b) || !mxIsSparse(C)) mexErrMsgIdAndTxt("cuadmm:sparse", "At, b and C must be sparse");
  const double sig = mxGetScalar(prhs[10]);
  double optional5[5] = {500, 50, 100, 1.1e4, 1.0};
  for (int i = 0; i < 5 && 11 + i < nrhs; ++i) optional5[i] = mxGetScalar(prhs[11 + i]);

  cuadmm_mex_result* res = nullptr;
  const int rc = cuadmm_mex_call(
      eig_streams, max_iter, stop_tol,
      mxGetM(At), mxGetN(At), (const size_t*)mxGetJc(At), (const size_t*)mxGetIr(At), mxGetPr(At),
      mxGetM(b), (const size_t*)mxGetJc(b), (const size_t*)mxGetIr(b), mxGetPr(b),
      mxGetM(C), (const size_t*)mxGetJc(C), (const size_t*)mxGetIr(C), mxGetPr(C),
      mxGetM(blk), mxGetPr(blk),
      mxGetM(X0), mxGetPr(X0), mxGetM(y0), mxGetPr(y0), mxGetM(S0), mxGetPr(S0),
      sig, nlhs, optional5, &res);
  if (rc != CUADMM_OK) mexErrMsgIdAndTxt("cuadmm:solve", "cuadmm_amd error %d: %s", rc, cuadmm_last_error());

  int vec_len = 0, con_num = 0, iter_num = 0;
  double total_time = 0;
  cuadmm_mex_result_dims(res, &vec_len, &con_num, &iter_num, &total_time);
  mxArray* mX = mxCreateDoubleMatrix(vec_len, 1, mxREAL);
  mxArray* my = mxCreateDoubleMatrix(con_num, 1, mxREAL);
  mxArray* mS = mxCreateDoubleMatrix(vec_len, 1, mxREAL);
  cuadmm_mex_result_XyS(res, mxGetPr(mX), mxGetPr(my), mxGetPr(mS));
  mxArray* info = mxCreateCellMatrix(10, 2);
  for (int r = 0; r < 10; ++r) mxSetCell(info, r, mxCreateString(kInfoNames[r]));
  mxSetCell(info, 0 + 10, mxCreateDoubleScalar((double)iter_num));
  for (int w = 0; w < 8; ++w) {             // CUADMM_INFO_* order = rows 1..8 of the cell
    mxArray* a = mxCreateDoubleMatrix(iter_num, 1, mxREAL);
    cuadmm_mex_result_info(res, w, mxGetPr(a));
    mxSetCell(info, (w + 1) + 10, a);
  }
  mxSetCell(info, 9 + 10, mxCreateDoubleScalar(total_time));
  cuadmm_mex_result_free(res);
  if (nlhs > 0) plhs[0] = mX; else mxDestroyArray(mX);
  if (nlhs > 1) plhs[1] = my; else mxDestroyArray(my);
  if (nlhs > 2) plhs[2] = mS; else mxDestroyArray(mS);
  if (nlhs > 3) plhs[3] = info; else mxDestroyArray(info);
}
