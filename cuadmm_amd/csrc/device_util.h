// HIP error plumbing shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

#define CUADMM_HIP_TRY(expr)                                                                      \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      ::cuadmm::set_error("HIP error %d (%s) at %s:%d: %s", (int)_e, hipGetErrorString(_e),       \
                          __FILE__, __LINE__, #expr);                                             \
      return CUADMM_ERR_NO_DEVICE;                                                                \
    }                                                                                             \
  } while (0)

namespace cuadmm {
// Blocking copies between PAGEABLE host memory and the device through the library's own page-locked staging buffers
// (staging.hip explains why the runtime must never get to register caller memory).  `after`: a stream to drain first.
int staged_h2d(void* dst, const void* src, size_t bytes, hipStream_t after = nullptr);
int staged_d2h(void* dst, const void* src, size_t bytes, hipStream_t after = nullptr);
int staged_h2d_2d(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width_bytes, size_t rows, hipStream_t after = nullptr);

constexpr size_t kMaxLdsBytes = 160 * 1024;   // gfx950: 160 KiB LDS per CU / per workgroup
constexpr int kMaxBlockSize = 4000;           // single-workgroup HBM-resident path limit (5n doubles of LDS)
// The RANK-LIMITED projection (option eig_rank) of a block beyond this size is refused unless option eig_allow_slow = 1: there
// one workgroup runs the whole QL iteration on a matrix in HBM -- measured 3.2 s at n = 1024 and 76 s at n = 2000.  The explicit
// eigendecomposition op (cuadmm_op_batch_eig) runs large matrices on the whole chip instead (eig_large.hip, up to n = 8192), and
// the solver's ordinary projection needs neither (matrix-sign path).
constexpr int kMaxEigSize = 1024;
}  // namespace cuadmm
