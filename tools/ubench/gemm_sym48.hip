// Micro-benchmark for BASELINE configs[2] (one block of n = 2000): the mirrored fp64-MFMA product of psd_large.hip
// (lg_gemm_sym_kernel: upper-triangle tiles of C = A B, A symmetric, operands staged global -> registers -> LDS, k-tiles of 32)
// with three tilings of the output at N = 2016 and N = 1056 / 1024:
//   32 x 32, four wavefronts of one 16 x 16 MFMA tile each  (the shipped kernel: 2 080 tiles at N = 2016, 6 workgroups per CU
//            = 1 536 slots -> 1.35 rounds; two LDS fragment reads per MFMA)
//   48 x 48, three wavefronts of a 16 x 48 strip each       (903 tiles at N = 2016: ONE round on 1 024+ slots; the A fragment is
//            shared by three MFMAs: 1.33 LDS reads per MFMA, 1.5 x the flops per operand byte)
//   64 x 64, four wavefronts of 32 x 32 each                (528 tiles: 2.06 per CU)
// Only the k-loop and a plain store of the tile are timed (the mirror / statistics epilogue is the same work for all three).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/gemm_sym48.hip -o /tmp/gemm_sym48.exe && /tmp/gemm_sym48.exe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void tri_decode(int t, int& bx, int& by) {   // t -> (bx >= by), t = bx (bx + 1) / 2 + by
  int b = (int)((__builtin_sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while (b * (b + 1) / 2 > t) --b;
  while ((b + 1) * (b + 2) / 2 <= t) ++b;
  bx = b; by = t - b * (b + 1) / 2;
}

template <int TM, int NWY, int NWX, int BK>
__global__ __launch_bounds__(64 * NWY * NWX) void gemm_sym_kernel(int N, const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C) {
  constexpr int NTHR = 64 * NWY * NWX, LDS = TM + 16, WR = TM / NWY, WC = TM / NWX, NTR = WR / 16, NTC = WC / 16;
  constexpr int TPR = NTHR / BK, PT = TM / TPR, NV = PT / 2;      // threads per k-row, doubles per thread, double2 per thread
  static_assert(TM % TPR == 0 && PT % 2 == 0, "staging layout");
  __shared__ double smem[2 * BK * LDS];
  double* As = smem;
  double* Bs = smem + BK * LDS;
  int bx, by;
  {   // the XCD-aware order of psd_large.hip: workgroup i runs on XCD i % 8; every XCD walks a contiguous range of 8 x 8-tile super-blocks
    const int grid_x = (int)gridDim.x, tile_x = (int)blockIdx.x;
    const int per_xcd = grid_x / 8;
    const int L = (tile_x % 8) * per_xcd + tile_x / 8;
    int sbx, sby;
    tri_decode(L / 64, sbx, sby);
    by = sby * 8 + (L % 64) / 8;
    bx = sbx * 8 + (L % 64) % 8;
    if (bx < by || bx >= N / TM) return;
  }
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wy = wave / NWX, wx = wave % NWX;
  const int row0 = by * TM, col0 = bx * TM;
  const int r16 = lane & 15, kk = lane >> 4;
  v4f64 acc[NTR][NTC];
#pragma unroll
  for (int i = 0; i < NTR; ++i)
#pragma unroll
    for (int j = 0; j < NTC; ++j) acc[i][j] = v4f64{0, 0, 0, 0};
  const int lk = tid / TPR, lc = (tid % TPR) * PT;
  const double2* ap = reinterpret_cast<const double2*>(A + (size_t)lk * N + row0 + lc);
  const double2* bp = reinterpret_cast<const double2*>(B + (size_t)lk * N + col0 + lc);
  const size_t kstep = (size_t)BK * N / 2;
  // staging registers as scalars (arrays end up in scratch)
  static_assert(NV <= 4, "staging registers");
  double2 Pa0 = {0, 0}, Pa1 = {0, 0}, Pa2 = {0, 0}, Pa3 = {0, 0}, Pb0 = {0, 0}, Pb1 = {0, 0}, Pb2 = {0, 0}, Pb3 = {0, 0};
  double2 Qa0 = {0, 0}, Qa1 = {0, 0}, Qa2 = {0, 0}, Qa3 = {0, 0}, Qb0 = {0, 0}, Qb1 = {0, 0}, Qb2 = {0, 0}, Qb3 = {0, 0};
#define LOADR(r) do { r##a0 = ap[0]; r##b0 = bp[0]; if constexpr (NV > 1) { r##a1 = ap[1]; r##b1 = bp[1]; } if constexpr (NV > 2) { r##a2 = ap[2]; r##b2 = bp[2]; } if constexpr (NV > 3) { r##a3 = ap[3]; r##b3 = bp[3]; } } while (0)
#define STORER(r) do { sa[0] = r##a0; sb[0] = r##b0; if constexpr (NV > 1) { sa[1] = r##a1; sb[1] = r##b1; } if constexpr (NV > 2) { sa[2] = r##a2; sb[2] = r##b2; } if constexpr (NV > 3) { sa[3] = r##a3; sb[3] = r##b3; } } while (0)
  LOADR(P);
  ap += kstep; bp += kstep;
  LOADR(Q);
  double2* sa = reinterpret_cast<double2*>(As + lk * LDS + lc);
  double2* sb = reinterpret_cast<double2*>(Bs + lk * LDS + lc);
#define COMPUTE()                                                                                               \
  _Pragma("unroll") for (int ks = 0; ks < BK; ks += 4) {                                                        \
    double af[NTR], bf[NTC];                                                                                      \
    _Pragma("unroll") for (int t = 0; t < NTR; ++t) af[t] = As[(ks + kk) * LDS + wy * WR + t * 16 + r16];         \
    _Pragma("unroll") for (int t = 0; t < NTC; ++t) bf[t] = Bs[(ks + kk) * LDS + wx * WC + t * 16 + r16];         \
    _Pragma("unroll") for (int i = 0; i < NTR; ++i)                                                               \
      _Pragma("unroll") for (int j = 0; j < NTC; ++j)                                                             \
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);                       \
  }
  for (int k0 = 0; k0 < N; k0 += 2 * BK) {
    __syncthreads();
    STORER(P);
    __syncthreads();
    if (k0 + 2 * BK < N) { ap += kstep; bp += kstep; LOADR(P); }
    COMPUTE();
    if (k0 + BK >= N) break;
    __syncthreads();
    STORER(Q);
    __syncthreads();
    if (k0 + 3 * BK < N) { ap += kstep; bp += kstep; LOADR(Q); }
    COMPUTE();
  }
#pragma unroll
  for (int i = 0; i < NTR; ++i)
#pragma unroll
    for (int j = 0; j < NTC; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) C[(size_t)(row0 + wy * WR + i * 16 + kk + 4 * r) * N + col0 + wx * WC + j * 16 + r16] = acc[i][j][r];
}

// 32 x 32 tiles with LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write, ONE barrier per k-tile, two LDS buffers.
// The instruction writes wave-uniform base + lane x 16 B, so the tile is unpadded; the four k-rows of a fragment read are kept on
// disjoint banks by swizzling the SOURCE column with the row's parity (col ^ 16 for odd k-rows) and reading with the same swizzle.
__global__ __launch_bounds__(256) void gemm_sym_glds_kernel(int N, const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C) {
  constexpr int TM = 32, BK = 32;
  __shared__ double smem[4 * BK * TM];                     // A0 | B0 | A1 | B1
  int bx, by;
  {
    const int grid_x = (int)gridDim.x, tile_x = (int)blockIdx.x;
    const int per_xcd = grid_x / 8;
    const int L = (tile_x % 8) * per_xcd + tile_x / 8;
    int sbx, sby;
    tri_decode(L / 64, sbx, sby);
    by = sby * 8 + (L % 64) / 8;
    bx = sbx * 8 + (L % 64) % 8;
    if (bx < by || bx >= N / TM) return;
  }
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wy = wave >> 1, wx = wave & 1;
  const int row0 = by * TM, col0 = bx * TM;
  const int r16 = lane & 15, kk = lane >> 4;
  v4f64 acc = {0, 0, 0, 0};
  // this lane's part of a chunk: k-row 4 c + (lane >> 4), 16-byte unit lane & 15 of the row, source column swizzled by the row parity
  const int kr_in = lane >> 4, p16 = lane & 15;
  auto issue = [&](int k0, int buf) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int c = 2 * wave + q, kr = 4 * c + kr_in;
      const int col = (2 * p16) ^ (16 * (kr & 1));
      const double* ga = A + (size_t)(k0 + kr) * N + row0 + col;
      const double* gb = B + (size_t)(k0 + kr) * N + col0 + col;
      double* la = smem + buf * 2 * BK * TM + c * 128;
      double* lb = smem + buf * 2 * BK * TM + BK * TM + c * 128;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga, (__attribute__((address_space(3))) void*)la, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gb, (__attribute__((address_space(3))) void*)lb, 16, 0, 0);
    }
  };
  issue(0, 0);
  const int nkt = N / BK;
  for (int kt = 0; kt < nkt; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nkt) issue((kt + 1) * BK, (kt + 1) & 1);
    const double* As = smem + (kt & 1) * 2 * BK * TM;
    const double* Bs = As + BK * TM;
#pragma unroll
    for (int ks = 0; ks < BK; ks += 4) {
      const int sw = 16 * ((ks + kk) & 1);
      const double af = As[(ks + kk) * TM + ((wy * 16 + r16) ^ sw)];
      const double bf = Bs[(ks + kk) * TM + ((wx * 16 + r16) ^ sw)];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) C[(size_t)(row0 + wy * 16 + kk + 4 * r) * N + col0 + wx * 16 + r16] = acc[r];
}

// Round 5: the same 32 x 32 tiling with LDS-DMA as the guide prescribes it for a DMA that has to survive a barrier
// (cdna_hip_programming.md, "Pipelining across barriers"): THREE LDS buffers, raw s_barrier + lgkmcnt(0) only (a __syncthreads()
// fence waits vmcnt(0) and drains the DMA queue), counted vmcnt so that k-tile t + 2 stays in flight while t is multiplied:
//   s_waitcnt vmcnt(NPT) -> tile t has landed, tile t + 1 stays in flight across the barrier ; s_barrier ; issue(t + 2) ; compute(t)
// (NPT = DMA instructions per thread and k-tile = 4).  PERSIST: workgroups take tiles from an atomic counter (longest-first order
// is the tile order itself: all tiles are equal), so a workgroup slot that finishes early takes the next tile instead of idling
// through the 35 %-full second round of a static grid.
template <bool PERSIST>
__global__ __launch_bounds__(256) void gemm_sym_glds3_kernel(int N, const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, int* counter, int ntile_slots) {
  constexpr int TM = 32, BK = 32, NBUF = 3;
  __shared__ double smem[NBUF * 2 * BK * TM];              // (A | B) x 3: 48 KB -> three workgroups per CU
  __shared__ int next_tile;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wy = wave >> 1, wx = wave & 1;
  const int r16 = lane & 15, kk = lane >> 4;
  const int kr_in = lane >> 4, p16 = lane & 15;
  const int nkt = N / BK;
  int tile_x = (int)blockIdx.x;
  for (;;) {
    if (PERSIST) {
      if (tid == 0) next_tile = atomicAdd(counter, 1);
      __syncthreads();
      tile_x = next_tile;
      __syncthreads();
      if (tile_x >= ntile_slots) return;
    }
    int bx, by;
    {
      const int grid_x = ntile_slots;
      const int per_xcd = grid_x / 8;
      const int L = (tile_x % 8) * per_xcd + tile_x / 8;
      int sbx, sby;
      tri_decode(L / 64, sbx, sby);
      by = sby * 8 + (L % 64) / 8;
      bx = sbx * 8 + (L % 64) % 8;
    }
    if (!(bx < by || bx >= N / TM)) {
      const int row0 = by * TM, col0 = bx * TM;
      v4f64 acc = {0, 0, 0, 0};
      auto issue = [&](int k0, int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int c = 2 * wave + q, kr = 4 * c + kr_in;
          const int col = (2 * p16) ^ (16 * (kr & 1));
          const double* ga = A + (size_t)(k0 + kr) * N + row0 + col;
          const double* gb = B + (size_t)(k0 + kr) * N + col0 + col;
          double* la = smem + buf * 2 * BK * TM + c * 128;
          double* lb = smem + buf * 2 * BK * TM + BK * TM + c * 128;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga, (__attribute__((address_space(3))) void*)la, 16, 0, 0);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gb, (__attribute__((address_space(3))) void*)lb, 16, 0, 0);
        }
      };
      issue(0, 0);
      issue(BK, 1);
      for (int kt = 0; kt < nkt; ++kt) {
        // this wavefront's part of tile kt has landed; tile kt + 1 (4 DMAs) stays in flight ACROSS the barrier
        if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                            // tile kt complete; everybody is done reading tile kt - 1 ...
        if (kt + 2 < nkt) issue((kt + 2) * BK, (kt + 2) % NBUF); // ... whose buffer takes tile kt + 2
        const double* As = smem + (kt % NBUF) * 2 * BK * TM;
        const double* Bs = As + BK * TM;
#pragma unroll
        for (int ks = 0; ks < BK; ks += 4) {
          const int sw = 16 * ((ks + kk) & 1);
          const double af = As[(ks + kk) * TM + ((wy * 16 + r16) ^ sw)];
          const double bf = Bs[(ks + kk) * TM + ((wx * 16 + r16) ^ sw)];
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af, bf, acc, 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the fragment reads of tile kt are done before this wavefront signals barrier kt + 1
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) C[(size_t)(row0 + wy * 16 + kk + 4 * r) * N + col0 + wx * 16 + r16] = acc[r];
    }
    if (!PERSIST) return;
  }
}

template <bool PERSIST>
static void run_glds3(int N, const double* A, const double* B, double* C) {
  const int nb = N / 32, sb = (nb + 7) / 8, tiles = sb * (sb + 1) / 2 * 64;
  int* counter; hipMalloc(&counter, sizeof(int));
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int grid = PERSIST ? p.multiProcessorCount * 3 : tiles;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    for (int q = 0; q < 10; ++q) {
      if (PERSIST) hipMemsetAsync(counter, 0, sizeof(int), 0);
      hipLaunchKernelGGL(gemm_sym_glds3_kernel<PERSIST>, dim3(grid), dim3(256), 0, 0, N, A, B, C, counter, tiles);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flops = 2.0 * N * 32.0 * 32.0 * (nb * (nb + 1) / 2);
  printf("  N = %4d  %-34s %5d tiles: %7.1f us per product, %5.1f TFLOP/s (upper-triangle tiles)\n", N,
         PERSIST ? "32 x 32, LDS-DMA x3, raw barrier, persistent" : "32 x 32, LDS-DMA x3, raw barrier", nb * (nb + 1) / 2, best * 100.0, flops / (best * 1e-4) / 1e12);
  hipFree(counter);
}

static void run_glds(int N, const double* A, const double* B, double* C) {
  const int nb = N / 32, sb = (nb + 7) / 8, tiles = sb * (sb + 1) / 2 * 64;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    for (int q = 0; q < 10; ++q) hipLaunchKernelGGL(gemm_sym_glds_kernel, dim3(tiles), dim3(256), 0, 0, N, A, B, C);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flops = 2.0 * N * 32.0 * 32.0 * (nb * (nb + 1) / 2);
  printf("  N = %4d  %-34s %5d tiles: %7.1f us per product, %5.1f TFLOP/s (upper-triangle tiles)\n", N, "32 x 32, LDS-DMA, 1 barrier / k-tile", nb * (nb + 1) / 2, best * 100.0, flops / (best * 1e-4) / 1e12);
}

template <int TM, int NWY, int NWX, int BK>
static void run(const char* name, int N, const double* A, const double* B, double* C) {
  const int nb = N / TM, sb = (nb + 7) / 8, tiles = sb * (sb + 1) / 2 * 64;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    for (int q = 0; q < 10; ++q) hipLaunchKernelGGL((gemm_sym_kernel<TM, NWY, NWX, BK>), dim3(tiles), dim3(64 * NWY * NWX), 0, 0, N, A, B, C);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flops = 2.0 * N * (double)TM * TM * (nb * (nb + 1) / 2);
  printf("  N = %4d  %-34s %5d tiles: %7.1f us per product, %5.1f TFLOP/s (upper-triangle tiles)\n", N, name, nb * (nb + 1) / 2, best * 100.0, flops / (best * 1e-4) / 1e12);
}

int main() {
  const int NMAX = 2112;
  double *A, *B, *C;
  hipMalloc(&A, sizeof(double) * NMAX * NMAX); hipMalloc(&B, sizeof(double) * NMAX * NMAX); hipMalloc(&C, sizeof(double) * NMAX * NMAX);
  std::vector<double> h((size_t)NMAX * NMAX);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) * 1e-3 - 0.5;
  hipMemcpy(A, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
  hipMemcpy(B, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice);
  std::vector<double> cref((size_t)NMAX * NMAX), cgot((size_t)NMAX * NMAX);
  auto compare = [&](int N, const char* what) {          // upper-triangle tiles against the shipped kernel's result, bit for bit (same k order)
    hipMemcpy(cgot.data(), C, sizeof(double) * (size_t)N * N, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < N; ++i)
      for (int j = (i / 32) * 32; j < N; ++j) { const double d = cgot[(size_t)i * N + j] - cref[(size_t)i * N + j]; worst = d > worst ? d : (-d > worst ? -d : worst); }
    printf("        %s vs shipped: max |difference| %.3e\n", what, worst);
  };
  for (int N : {2016, 2112}) {
    if (N % 32 == 0) { run<32, 2, 2, 32>("32 x 32, 4 waves (shipped)", N, A, B, C); hipMemcpy(cref.data(), C, sizeof(double) * (size_t)N * N, hipMemcpyDeviceToHost); }
    if (N % 32 == 0) run_glds(N, A, B, C);
    if (N % 32 == 0) { hipMemset(C, 0, sizeof(double) * (size_t)N * N); run_glds3<false>(N, A, B, C); compare(N, "LDS-DMA x3"); }
    if (N % 32 == 0) { hipMemset(C, 0, sizeof(double) * (size_t)N * N); run_glds3<true>(N, A, B, C); compare(N, "LDS-DMA x3 persistent"); }
    if (N % 48 == 0) run<48, 3, 1, 32>("48 x 48, 3 waves of 16 x 48", N, A, B, C);
    if (N % 48 == 0) run<48, 1, 3, 32>("48 x 48, 3 waves of 48 x 16", N, A, B, C);
    if (N % 64 == 0) run<64, 2, 2, 16>("64 x 64, 4 waves of 32 x 32", N, A, B, C);
    if (N % 96 == 0) run<96, 3, 2, 16>("96 x 96, 6 waves of 32 x 48", N, A, B, C);
    if (N % 64 == 0) run<64, 4, 1, 32>("64 x 64, 4 waves of 16 x 64", N, A, B, C);
  }
  for (int N : {1056, 1024, 1152}) {
    if (N % 32 == 0) run<32, 2, 2, 32>("32 x 32, 4 waves (shipped)", N, A, B, C);
    if (N % 32 == 0) run_glds(N, A, B, C);
    if (N % 48 == 0) run<48, 3, 1, 32>("48 x 48, 3 waves of 16 x 48", N, A, B, C);
    if (N % 64 == 0) run<64, 2, 2, 16>("64 x 64, 4 waves of 32 x 32", N, A, B, C);
  }
  return 0;
}
