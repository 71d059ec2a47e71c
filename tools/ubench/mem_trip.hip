// Micro-benchmark: how long does ONE wavefront wait for a burst of block-sized loads (the "trip 2" of psd_sign_closed.h: X and C
// ranges of a 32 x 32 block + its record, ~10 KB in ~30 load instructions) when the chip is populated like the persistent C2
// kernel -- 16 wavefronts per CU, each of them loading only ~10 % of the time?  Variants separate the suspects:
//   layout 0: three arrays of nblk x 4224 B (the engine's layout: X, C, rec far apart)        -> pages touched per trip: ~6
//   layout 1: one array, the three ranges of a block contiguous (AoS)                          -> ~3
//   nblk small (arrays of 2 MB each): everything TLB- and cache-resident
//   duty: ticks of s_sleep between trips (0: every wave loads all the time -> bandwidth-bound)
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mem_trip.hip -o tools/ubench/mem_trip.exe && tools/ubench/mem_trip.exe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int BLK = 528;        // doubles per block range (32 x 33 / 2)
constexpr int REC = 224;        // doubles per record (1792 B)

template <int NLOADS>
__global__ __launch_bounds__(1024) void trip_kernel(const double* __restrict__ X, const double* __restrict__ C, const double* __restrict__ R,
                                                    long long strideX, long long strideR, int nblk, int trips, int sleep_units, int write_back,
                                                    double* __restrict__ W, long long* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = blockIdx.x, G = gridDim.x;
  const int nb = (nblk - g + G - 1) / G;
  long long tot = 0, mx = 0;
  double acc = 0;
  unsigned seed = 12345u + 977u * (unsigned)(g * 16 + wave);
  // de-phase the waves
  for (int i = 0; i < (int)(seed % 97u); ++i) __builtin_amdgcn_s_sleep(32);
  for (int t = 0; t < trips; ++t) {
    seed = seed * 1664525u + 1013904223u;
    const int j = (int)((seed >> 8) % (unsigned)nb);
    const long long blk = g + (long long)j * G;
    const double* x = X + blk * strideX + lane;
    const double* c = C + blk * strideX + lane;
    const double* r = R + blk * strideR + lane;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long t0 = __builtin_readcyclecounter();
    double v[NLOADS];
#pragma unroll
    for (int u = 0; u < 9; ++u) if (u < NLOADS) v[u] = x[64 * u];
#pragma unroll
    for (int u = 0; u < 9; ++u) if (9 + u < NLOADS) v[9 + u] = c[64 * u];
#pragma unroll
    for (int u = 0; u < 3; ++u) if (18 + u < NLOADS) v[18 + u] = r[64 * u];
    double s = 0;
#pragma unroll
    for (int u = 0; u < NLOADS; ++u) s += v[u];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_readcyclecounter();
    acc += s;
    const long long d = t1 - t0;
    tot += d; mx = d > mx ? d : mx;
    if (write_back) {   // the task's stores: S and X ranges
      double* w = W + blk * strideX + lane;
#pragma unroll
      for (int u = 0; u < 8; ++u) { w[64 * u] = s + u; }
      double* w2 = W + (blk + nblk) * strideX + lane;
#pragma unroll
      for (int u = 0; u < 8; ++u) { w2[64 * u] = s - u; }
    }
    for (int i = 0; i < sleep_units; ++i) __builtin_amdgcn_s_sleep(127);
  }
  if (lane == 0) { out[2 * (g * 16 + wave)] = tot; out[2 * (g * 16 + wave) + 1] = mx; }
  if (acc == 1.2345e-300) W[0] = acc;
}

int main(int argc, char** argv) {
  const int G = 256, WAVES = 16;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
  long long* out;
  CHECK(hipMalloc(&out, sizeof(long long) * 2 * G * WAVES));
  std::vector<long long> h(2 * G * WAVES);
  struct Cfg { const char* name; int nblk; int layout; int sleep; int wb; int nloads; };
  const Cfg cfgs[] = {
      {"engine layout, 10000 blk, duty ~10%, +stores, 21 loads", 10000, 0, 14, 1, 21},
      {"engine layout, 10000 blk, duty ~10%, no stores, 21 loads", 10000, 0, 14, 0, 21},
      {"engine layout, 10000 blk, duty ~10%, no stores, 9 loads (X only)", 10000, 0, 14, 0, 9},
      {"AoS layout,    10000 blk, duty ~10%, +stores, 21 loads", 10000, 1, 14, 1, 21},
      {"AoS layout,    10000 blk, duty ~10%, no stores, 21 loads", 10000, 1, 14, 0, 21},
      {"engine layout,   480 blk (2 MB arrays), duty ~10%, no stores", 480, 0, 14, 0, 21},
      {"engine layout, 10000 blk, one wave per CU loads (others absent)", 10000, 0, 14, 0, 21},
      {"engine layout, 10000 blk, every wave loads all the time, no stores", 10000, 0, 0, 0, 21},
      {"engine layout, 100000 blk, duty ~10%, no stores", 100000, 0, 14, 0, 21},
  };
  int ci = 0;
  for (const Cfg& cf : cfgs) {
    const long long strideX = cf.layout == 0 ? BLK : (2 * BLK + REC);
    const long long strideR = cf.layout == 0 ? REC : (2 * BLK + REC);
    const size_t nX = (size_t)cf.nblk * strideX + 4096;
    double *X, *C, *R, *W;
    if (cf.layout == 0) {
      CHECK(hipMalloc(&X, nX * 8)); CHECK(hipMalloc(&C, nX * 8)); CHECK(hipMalloc(&R, ((size_t)cf.nblk * REC + 4096) * 8));
    } else {
      CHECK(hipMalloc(&X, nX * 8)); C = X + BLK; R = X + 2 * BLK;
    }
    CHECK(hipMalloc(&W, (2 * (size_t)cf.nblk * strideX + 8192) * 8));
    CHECK(hipMemset(X, 0, nX * 8));
    if (cf.layout == 0) { CHECK(hipMemset(C, 0, nX * 8)); CHECK(hipMemset(R, 0, ((size_t)cf.nblk * REC + 4096) * 8)); }
    const int trips = 200;
    const int threads = ci == 6 ? 64 : 64 * WAVES;
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipMemset(out, 0, sizeof(long long) * 2 * G * WAVES));
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      CHECK(hipEventRecord(e0));
      if (cf.nloads == 9) hipLaunchKernelGGL(trip_kernel<9>, dim3(G), dim3(threads), 0, 0, X, C, R, strideX, strideR, cf.nblk, trips, cf.sleep, cf.wb, W, out);
      else hipLaunchKernelGGL(trip_kernel<21>, dim3(G), dim3(threads), 0, 0, X, C, R, strideX, strideR, cf.nblk, trips, cf.sleep, cf.wb, W, out);
      CHECK(hipEventRecord(e1));
      CHECK(hipDeviceSynchronize());
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      CHECK(hipMemcpy(h.data(), out, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
      double tot = 0; long long mx = 0; int nw = 0;
      for (int w = 0; w < G * WAVES; ++w) if (h[2 * w] > 0) { tot += (double)h[2 * w]; mx = std::max(mx, h[2 * w + 1]); ++nw; }
      if (rep == 1)
        printf("%-72s: %7.0f ticks per trip (max %lld), %d waves, kernel %.2f ms, %.2f TB/s read\n", cf.name, tot / nw / trips, mx, nw, ms,
               (double)nw * trips * cf.nloads * 512.0 / (ms * 1e-3) * 1e-12);
    }
    CHECK(hipFree(X)); if (cf.layout == 0) { CHECK(hipFree(C)); CHECK(hipFree(R)); } CHECK(hipFree(W));
    ++ci;
  }
  return 0;
}
