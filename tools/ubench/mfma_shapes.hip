// Micro-benchmark: issue cost of the two fp64 MFMA shapes of gfx950 -- v_mfma_f64_16x16x4_f64 (2048 flop, one 16 x 16 tile) and
// v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4 x 4 x 4: 512 flop) -- with WPS wavefronts per SIMD issuing independent chains.  Question
// behind it (round 5): would a symmetric product at 4 x 4 granularity (36 of 64 sub-blocks of a 32 x 32 tile instead of 3 of 4
// 16 x 16 sub-tiles: -25 % flops) pay?  Only if the small shape keeps the big one's flop rate.
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_shapes.hip -o tools/ubench/mfma_shapes.exe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4f64 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void k_big(double* out, long long* ticks, int iters) {
  v4f64 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0;
  const double a = out[threadIdx.x] + 1e-3, b = a + 1e-3;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, a3, 0, 0, 0);
      a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, a4, 0, 0, 0);
      a5 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, a5, 0, 0, 0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  const v4f64 s = a0 + a1 + a2 + a3 + a4 + a5;
  if (s[0] + s[1] + s[2] + s[3] == 12345.678) out[threadIdx.x] = s[0];
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

__global__ __launch_bounds__(64) void k_small(double* out, long long* ticks, int iters) {
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
  const double a = out[threadIdx.x] + 1e-3, b = a + 1e-3;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, a3, 0, 0, 0);
      a4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, a, a4, 0, 0, 0);
      a5 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, b, a5, 0, 0, 0);
      a6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, a6, 0, 0, 0);
      a7 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, a, a7, 0, 0, 0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  const double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (s == 12345.678) out[threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double* out; long long* ticks;
  hipMalloc(&out, 64 * sizeof(double)); hipMemset(out, 0, 64 * sizeof(double));
  hipMalloc(&ticks, cus * 4 * 8 * sizeof(long long));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%s: %d CUs\n", p.gcnArchName, cus);
  for (int shape = 0; shape < 2; ++shape)
    for (int wps : {1, 2, 4}) {
      const int waves = cus * 4 * wps, iters = 20000;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (shape == 0) hipLaunchKernelGGL(k_big, dim3(waves), dim3(64), 0, 0, out, ticks, iters);
        else hipLaunchKernelGGL(k_small, dim3(waves), dim3(64), 0, 0, out, ticks, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long t; hipMemcpy(&t, ticks, sizeof t, hipMemcpyDeviceToHost);
        const double per = 24.0, flop1 = shape == 0 ? 2048.0 : 512.0;
        if (rep == 1)
          printf("  %-28s %d waves/SIMD: %8.3f ms  %6.1f TFLOP/s  %.1f ticks per MFMA issue slot of a SIMD\n",
                 shape == 0 ? "v_mfma_f64_16x16x4_f64" : "v_mfma_f64_4x4x4_4b_f64", wps, ms, (double)waves * iters * per * flop1 / (ms * 1e-3) / 1e12,
                 (double)t / (iters * per * wps));
      }
    }
  return 0;
}
