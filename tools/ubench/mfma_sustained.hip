// Micro-benchmark: SUSTAINED fp64 matrix-core throughput of the whole chip (gfx950) -- every SIMD runs WPS wavefronts of
// independent v_mfma_f64_16x16x4_f64 chains with no memory traffic.  Prints TFLOP/s (hipEvent time) and the shader clock
// implied by s_memtime ticks per wall second, for launch lengths from ~0.1 ms to ~100 ms: what the matrix cores deliver
// under power management, i.e. the achievable ceiling the projection kernels are measured against (the paper peak is
// 256 CUs x 4 SIMDs x 2048 flop / 64 cycles x 2.4 GHz = 78.6 TFLOP/s).
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_sustained.hip -o tools/ubench/mfma_sustained.exe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4f64 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void k_mfma(double* out, long long* ticks, int iters) {
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

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 2;        // wavefronts per SIMD
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, waves = cus * 4 * wps;
  double* out; long long* ticks;
  hipMalloc(&out, 64 * sizeof(double)); hipMemset(out, 0, 64 * sizeof(double));
  hipMalloc(&ticks, waves * sizeof(long long));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%s: %d CUs, %d wavefronts (%d per SIMD), clockRate %d kHz\n", p.gcnArchName, cus, waves, wps, p.clockRate);
  for (int rep = 0; rep < 2; ++rep)
    for (int iters : {100, 1000, 10000, 100000}) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_mfma, dim3(waves), dim3(64), 0, 0, out, ticks, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long t; hipMemcpy(&t, ticks, sizeof t, hipMemcpyDeviceToHost);
      const double flop = (double)waves * iters * 24 * 2048.0;
      printf("  iters %6d: %9.3f ms  %6.1f TFLOP/s  | wave 0: %lld ticks, %.1f ticks per MFMA issue slot (x%d waves), ticks/s = %.2f GHz\n", iters, ms,
             flop / (ms * 1e-3) / 1e12, t, (double)t / (iters * 24.0 * wps), wps, (double)t / (ms * 1e-3) / 1e9);
    }
  return 0;
}
