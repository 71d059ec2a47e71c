// Micro-benchmark: dependent-chain latency and issue rate of fp64 VALU ops on gfx950 (one wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 4096
__global__ void k_fma_dep(double* out, double a, double b, long long* cyc) {
  double x = out[threadIdx.x];
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 64
  for (int i = 0; i < N; ++i) x = fma(x, a, b);
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_fma_indep(double* out, double a, double b, long long* cyc) {
  double x0 = out[threadIdx.x], x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
  for (int i = 0; i < N / 8; ++i) { x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b); x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b); }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_rsq_dep(double* out, long long* cyc) {
  double x = out[threadIdx.x];
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
  for (int i = 0; i < N / 4; ++i) x = __builtin_amdgcn_rsq(x) + 1.0;
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_mul_add_dep(double* out, double a, long long* cyc) {
  double x = out[threadIdx.x];
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 32
  for (int i = 0; i < N / 2; ++i) { x = x * a; x = x + a; }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_lds_dep(double* out, long long* cyc) {
  __shared__ double sm[64];
  sm[threadIdx.x] = (double)((threadIdx.x + 1) & 63);
  __syncthreads();
  int idx = threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N / 4; ++i) idx = (int)sm[idx];
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = idx; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_shfl_dep(double* out, long long* cyc) {
  double x = out[threadIdx.x];
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N / 4; ++i) x += __shfl_xor(x, 16, 64);
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
typedef double v4f64 __attribute__((ext_vector_type(4)));
__global__ void k_mfma_f64(double* out, long long* cyc) {
  v4f64 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  double a = out[threadIdx.x], b = a + 1;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N / 4; ++i) {
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3]; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_mfma_f64_dep(double* out, long long* cyc) {
  v4f64 acc0 = {0, 0, 0, 0};
  double a = out[threadIdx.x], b = a + 1;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N / 4; ++i) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = acc0[0]; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* d; long long* c; hipMalloc(&d, 64 * 8); hipMalloc(&c, 8);
  std::vector<double> h(64, 1.000001); long long hc;
  auto run = [&](const char* name, auto launch, int ops) {
    for (int rep = 0; rep < 2; ++rep) { hipMemcpy(d, h.data(), 64 * 8, hipMemcpyHostToDevice); launch(); hipDeviceSynchronize(); }
    hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    printf("%-28s %8lld cycles / %5d ops = %6.2f cycles/op\n", name, hc, ops, (double)hc / ops);
  };
  run("fma_f64 dependent", [&] { hipLaunchKernelGGL(k_fma_dep, 1, 64, 0, 0, d, 0.999999, 1e-7, c); }, N);
  run("fma_f64 8 independent", [&] { hipLaunchKernelGGL(k_fma_indep, 1, 64, 0, 0, d, 0.999999, 1e-7, c); }, N);
  run("mul_f64+add_f64 dependent", [&] { hipLaunchKernelGGL(k_mul_add_dep, 1, 64, 0, 0, d, 0.999999, c); }, N);
  run("rsq_f64+add dependent (pair)", [&] { hipLaunchKernelGGL(k_rsq_dep, 1, 64, 0, 0, d, c); }, N / 4);
  run("lds read dependent", [&] { hipLaunchKernelGGL(k_lds_dep, 1, 64, 0, 0, d, c); }, N / 4);
  run("shfl_xor(16) f64 + add dep", [&] { hipLaunchKernelGGL(k_shfl_dep, 1, 64, 0, 0, d, c); }, N / 4);
  run("mfma_f64_16x16x4 4 indep", [&] { hipLaunchKernelGGL(k_mfma_f64, 1, 64, 0, 0, d, c); }, N);
  run("mfma_f64_16x16x4 dependent", [&] { hipLaunchKernelGGL(k_mfma_f64_dep, 1, 64, 0, 0, d, c); }, N / 4);
  return 0;
}
