// Micro-benchmark: the shader clock the chip actually runs at under fp64 matrix-core load (gfx950).  Every wavefront
// reads s_memtime (shader-clock ticks) and s_memrealtime (constant 100 MHz) around its loop: their ratio is the clock.
// Variants: pure MFMA chains | MFMA + LDS fragment reads + VALU (the mix of the sign kernels) | VALU only.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/clock_under_load.hip -o tools/ubench/clock_under_load.exe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(64) void k_load(double* out, long long* stamps, int iters) {
  __shared__ double lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 1e-3 * i;
  v4f64 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  double a = out[threadIdx.x] + 1e-3, b = a + 1e-3, acc = 0.0;
  const long long t0 = __builtin_readcyclecounter();
  const long long r0 = (long long)__builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE <= 1) {
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        if (MODE == 1) { a = lds[(threadIdx.x * 33 + u * 64 + i) & 2047]; b = lds[(threadIdx.x + u * 128 + i * 7) & 2047]; }
        a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, a3, 0, 0, 0);
        if (MODE == 1) { acc = fma(acc, 1.0000001, a0[0]); acc = fma(acc, 0.9999999, a1[1]); }
      }
    } else {
#pragma unroll
      for (int u = 0; u < 48; ++u) acc = fma(acc, 1.0000001, b);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  const long long r1 = (long long)__builtin_amdgcn_s_memrealtime();
  const v4f64 s = a0 + a1 + a2 + a3;
  if (s[0] + s[1] + s[2] + s[3] + acc == 12345.678) out[threadIdx.x] = s[0];
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 2;
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int waves = p.multiProcessorCount * 4 * wps;
  double* out; long long* st;
  (void)hipMalloc(&out, 64 * sizeof(double)); (void)hipMemset(out, 0, 64 * sizeof(double));
  (void)hipMalloc(&st, 2 * waves * sizeof(long long));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const char* names[3] = {"pure MFMA", "MFMA + LDS reads + VALU", "VALU only"};
  for (int mode = 0; mode < 3; ++mode)
    for (int iters : {2000, 20000}) {
      (void)hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k_load<0>, dim3(waves), dim3(64), 0, 0, out, st, iters);
      else if (mode == 1) hipLaunchKernelGGL(k_load<1>, dim3(waves), dim3(64), 0, 0, out, st, iters);
      else hipLaunchKernelGGL(k_load<2>, dim3(waves), dim3(64), 0, 0, out, st, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      long long h[2]; (void)hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost);
      const double flop = mode <= 1 ? (double)waves * iters * 24 * 2048.0 : 0.0;
      printf("%-26s %d waves/SIMD iters %6d: %8.3f ms  %6.1f TFLOP/s | wave 0: %lld s_memtime ticks, %lld s_memrealtime ticks (100 MHz) -> %.3f GHz\n",
             names[mode], wps, iters, ms, flop / (ms * 1e-3) / 1e12, h[0], h[1], h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0);
    }
  return 0;
}
