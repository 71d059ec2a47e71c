// Wave-wide deterministic sums on the DPP crossbar (shared by the matrix-sign kernels).
#pragma once
#include <hip/hip_runtime.h>

namespace cuadmm {

// ---------------------------------------------------------------------------------------------------------------
// Wave-wide sums for the schedule statistics (sign_sched.h): xor-butterfly inside each row of 16 lanes on the DPP
// crossbar (quad_perm, row_half_mirror, row_mirror -- no LDS traffic, fixed association order, so every lane of a row
// holds bit-identical sums), then the four row sums through v_readlane.  The result is wave-uniform (SGPR operands),
// so the schedule's branches are scalar.
// ---------------------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double sw_dpp(double v) {
  // bound_ctrl: every lane has a valid source under these controls, and the destination needs no initialisation (without it the
  // compiler zeroes both halves first: two v_mov per step -- VALU slots the fp64 matrix pipe pays for)
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sw_readlane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum(double v) {
  v += sw_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
  v += sw_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  v += sw_dpp<0x141>(v);   // row_half_mirror
  v += sw_dpp<0x140>(v);   // row_mirror
  return (sw_readlane(v, 0) + sw_readlane(v, 16)) + (sw_readlane(v, 32) + sw_readlane(v, 48));
}


}  // namespace cuadmm
