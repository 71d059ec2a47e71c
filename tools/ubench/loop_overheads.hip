// Micro-benchmark: per-iteration cost of (a) an LDS store + dependent-free load + waitcnt(0), (b) a data dependent
// divergent exit test, (c) both, in a loop carrying a short fp64 chain.  One wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 2048
__global__ void k_chain_only(double* out, double a, double b, long long* cyc) {
  double x = out[threadIdx.x];
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; ++i) { x = fma(x, a, b); x = fma(x, a, b); x = fma(x, a, b); x = fma(x, a, b); }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_chain_break(double* out, double a, double b, long long* cyc) {
  double x = out[threadIdx.x];
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < N; ++i) { x = fma(x, a, b); x = fma(x, a, b); x = fma(x, a, b); x = fma(x, a, b); if (x == 0.0) break; }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_chain_lds(double* out, double a, double b, long long* cyc) {
  __shared__ double sm[N + 64];
  double x = out[threadIdx.x];
  for (int i = threadIdx.x; i < N + 64; i += 64) sm[i] = 1.0;
  __syncthreads();
  double e = sm[N];
  long long t0 = __builtin_readcyclecounter();
  for (int i = N - 1; i >= 1; --i) {
    double en = sm[i - 1];                    // "prefetch" of an untouched entry
    x = fma(x, a, e); x = fma(x, a, b); x = fma(x, a, b); x = fma(x, a, b);
    if (threadIdx.x == 0) sm[i + 1] = x;      // writer lane publishes
    e = en;
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_chain_lds_break(double* out, double a, double b, long long* cyc) {
  __shared__ double sm[N + 64];
  double x = out[threadIdx.x];
  for (int i = threadIdx.x; i < N + 64; i += 64) sm[i] = 1.0;
  __syncthreads();
  double e = sm[N];
  long long t0 = __builtin_readcyclecounter();
  for (int i = N - 1; i >= 1; --i) {
    double en = sm[i - 1];
    x = fma(x, a, e); x = fma(x, a, b); x = fma(x, a, b); x = fma(x, a, b);
    if (x == 0.0) break;
    if (threadIdx.x == 0) sm[i + 1] = x;
    e = en;
  }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_divergent_trip(double* out, double a, double b, long long* cyc) {
  double x = out[threadIdx.x];
  int lim = N - (threadIdx.x & 7);            // lanes leave the loop at different iterations
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < lim; ++i) { x = fma(x, a, b); x = fma(x, a, b); x = fma(x, a, b); x = fma(x, a, b); }
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  double* d; long long* c; hipMalloc(&d, 64 * 8); hipMalloc(&c, 8);
  std::vector<double> h(64, 1.000001); long long hc;
  auto run = [&](const char* name, auto launch) {
    for (int rep = 0; rep < 2; ++rep) { hipMemcpy(d, h.data(), 64 * 8, hipMemcpyHostToDevice); launch(); hipDeviceSynchronize(); }
    hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
    printf("%-40s %8lld cycles / %d iters = %7.1f cycles/iter\n", name, hc, N, (double)hc / N);
  };
  run("4 dependent fma (uniform loop)", [&] { hipLaunchKernelGGL(k_chain_only, 1, 64, 0, 0, d, 0.999999, 1e-7, c); });
  run("4 fma + data-dependent break", [&] { hipLaunchKernelGGL(k_chain_break, 1, 64, 0, 0, d, 0.999999, 1e-7, c); });
  run("4 fma + lds prefetch + writer store", [&] { hipLaunchKernelGGL(k_chain_lds, 1, 64, 0, 0, d, 0.999999, 1e-7, c); });
  run("4 fma + lds + store + break", [&] { hipLaunchKernelGGL(k_chain_lds_break, 1, 64, 0, 0, d, 0.999999, 1e-7, c); });
  run("4 fma, divergent trip counts", [&] { hipLaunchKernelGGL(k_divergent_trip, 1, 64, 0, 0, d, 0.999999, 1e-7, c); });
  return 0;
}
