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
// Lifts a kernel's cap on dynamic LDS to everything its static LDS leaves.  The attribute is per kernel and process-wide, so callers
// never set "what this launch needs": a later, smaller request would lower it under another object's larger launches.
inline hipError_t allow_max_dynamic_lds(const void* kern) {
  hipFuncAttributes fa;
  hipError_t e = hipFuncGetAttributes(&fa, kern);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kMaxLdsBytes - fa.sharedSizeBytes));
}
// The same once per kernel AND DEVICE (a function's attributes belong to the device's copy of it; a process may hold solvers on
// several devices): `static LdsCapOnce once;` beside the launch, `once(kern)` before it.
struct LdsCapOnce {
  bool done[64] = {};
  hipError_t operator()(const void* kern) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return allow_max_dynamic_lds(kern);
    if (done[dev]) return hipSuccess;
    e = allow_max_dynamic_lds(kern);
    if (e == hipSuccess) done[dev] = true;
    return e;
  }
};
constexpr int kMaxBlockSize = 8192;           // largest block the projection plans accept (= the explicit eigensolver's limit, eig_large.h)
}  // namespace cuadmm
