// Where does the one-workgroup-per-CU symmetric product (lg_gemm_big_kernel, psd_large.hip) lose its time?  The k-loop of that kernel
// with the global loads and / or the MFMAs compiled out, and with 4 or 8 wavefronts.  hipcc --offload-arch=gfx950 -O3 gemm_big.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void tri_decode(int e, int& c, int& r) {
  c = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
  while ((c + 1) * (c + 2) / 2 <= e) ++c;
  while (c * (c + 1) / 2 > e) --c;
  r = e - c * (c + 1) / 2;
}

// WAVES = 8: two k-halves; WAVES = 4: every wavefront takes the whole stage
template <int TM, int WAVES, bool LOAD, bool MFMA, int AHEAD>
__global__ __launch_bounds__(WAVES * 64) void k(int N, const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C) {
  constexpr int BK = 32, LD = TM + 16, STAGE = 2 * BK * LD, NTW = TM / 32, WT = TM / 2, NT = WAVES * 64, NLD = BK * TM / 2 / NT;
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = WAVES == 8 ? wave >> 2 : 0, wy = (wave >> 1) & 1, wx = wave & 1, r16 = lane & 15, kk = lane >> 4;
  constexpr int KH = WAVES == 8 ? BK / 2 : BK;
  const int nbt = N / TM, ntiles = nbt * (nbt + 1) / 2, G = gridDim.x, per = G / 8;
  const int L0 = (blockIdx.x % 8) * per + blockIdx.x / 8;
  int goff[NLD], soff[NLD];
#pragma unroll
  for (int p = 0; p < NLD; ++p) { const int idx = tid + NT * p, row = idx / (TM / 2), c2 = idx % (TM / 2); goff[p] = row * N + 2 * c2; soff[p] = row * LD + 2 * c2; }
  const int nk = N / BK;
  for (int tile = L0; tile < ntiles; tile += G) {
    int bx, by; tri_decode(tile, bx, by);
    const int row0 = by * TM, col0 = bx * TM;
    v4 acc[NTW][NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[i][j] = v4{0, 0, 0, 0};
    const double* ap = A + row0; const double* bp = B + col0;
    double2 ra[NLD], rb[NLD];
#pragma unroll
    for (int p = 0; p < NLD; ++p) { ra[p] = *(const double2*)(ap + goff[p]); rb[p] = *(const double2*)(bp + goff[p]); }
#pragma unroll
    for (int p = 0; p < NLD; ++p) { *(double2*)(smem + soff[p]) = ra[p]; *(double2*)(smem + BK * LD + soff[p]) = rb[p]; }
    ap += (size_t)BK * N; bp += (size_t)BK * N;
#pragma unroll
    for (int p = 0; p < NLD; ++p) { ra[p] = *(const double2*)(ap + goff[p]); rb[p] = *(const double2*)(bp + goff[p]); }
    __syncthreads();
    for (int s = 0; s < nk; ++s) {
      const double* As = smem + (s & 1) * STAGE + (half * KH) * LD;
      const double* Bs = As + BK * LD;
      if (s + 1 < nk) {
        double* nx = smem + ((s + 1) & 1) * STAGE;
#pragma unroll
        for (int p = 0; p < NLD; ++p) { *(double2*)(nx + soff[p]) = ra[p]; *(double2*)(nx + BK * LD + soff[p]) = rb[p]; }
        if (LOAD && s + 2 < nk) {
          ap += (size_t)BK * N; bp += (size_t)BK * N;
#pragma unroll
          for (int p = 0; p < NLD; ++p) { ra[p] = *(const double2*)(ap + goff[p]); rb[p] = *(const double2*)(bp + goff[p]); }
        }
      }
#pragma unroll
      for (int ks = 0; ks < KH; ks += 4) {
        double af[NTW], bf[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) { af[t] = As[(ks + kk) * LD + wy * WT + t * 16 + r16]; bf[t] = Bs[(ks + kk) * LD + wx * WT + t * 16 + r16]; }
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
          for (int j = 0; j < NTW; ++j) {
            if (MFMA) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
            else acc[i][j][0] += af[i] * bf[j];
          }
      }
      __syncthreads();
    }
    if (half == 0) {
#pragma unroll
      for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = row0 + wy * WT + i * 16 + kk + 4 * r, col = col0 + wx * WT + j * 16 + r16;
            C[(size_t)row * N + col] = acc[i][j][r];
          }
    }
    __syncthreads();
  }
}

template <int TM, int WAVES, bool LOAD, bool MFMA, int AHEAD>
static int run(const char* name, int N, double* A, double* B, double* C) {
  constexpr int LD = TM + 16;
  const size_t lds = sizeof(double) * 2 * 2 * 32 * LD;
  auto kern = k<TM, WAVES, LOAD, MFMA, AHEAD>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int nb = N / TM, nt = nb * (nb + 1) / 2, grid = nt < 256 ? (nt + 7) / 8 * 8 : 256;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, 0, N, A, B, C);
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, 0, N, A, B, C);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1000.0 / reps, fl = 2.0 * TM * TM * (double)N * nt;
  printf("%-44s N=%d tiles=%d grid=%d  %.1f us  %.1f TFLOP/s\n", name, N, nt, grid, us, fl / us * 1e-6);
  return 0;
}

// two register sets: stages s + 2 and s + 3 in flight while stage s is multiplied
template <int TM, int PAD, bool LOAD, bool MFMA>
__global__ __launch_bounds__(512) void k2(int N, const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C) {
  constexpr int BK = 32, LD = TM + PAD, STAGE = 2 * BK * LD, NTW = TM / 32, WT = TM / 2, NT = 512, NLD = BK * TM / 2 / NT;
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = wave >> 2, wy = (wave >> 1) & 1, wx = wave & 1, r16 = lane & 15, kk = lane >> 4;
  constexpr int KH = BK / 2;
  const int nbt = N / TM, ntiles = nbt * (nbt + 1) / 2, G = gridDim.x, per = G / 8;
  const int L0 = (blockIdx.x % 8) * per + blockIdx.x / 8;
  int goff[NLD], soff[NLD];
#pragma unroll
  for (int p = 0; p < NLD; ++p) { const int idx = tid + NT * p, row = idx / (TM / 2), c2 = idx % (TM / 2); goff[p] = row * N + 2 * c2; soff[p] = row * LD + 2 * c2; }
  const int nk = N / BK;
  for (int tile = L0; tile < ntiles; tile += G) {
    int bx, by; tri_decode(tile, bx, by);
    const int row0 = by * TM, col0 = bx * TM;
    v4 acc[NTW][NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[i][j] = v4{0, 0, 0, 0};
    const double* ap = A + row0; const double* bp = B + col0;
    double2 xa[NLD], xb[NLD], ya[NLD], yb[NLD];
#define LOADSET(a, b) _Pragma("unroll") for (int p = 0; p < NLD; ++p) { a[p] = *(const double2*)(ap + goff[p]); b[p] = *(const double2*)(bp + goff[p]); } ap += (size_t)BK * N; bp += (size_t)BK * N;
#define STORESET(a, b, buf) _Pragma("unroll") for (int p = 0; p < NLD; ++p) { *(double2*)(smem + (buf) * STAGE + soff[p]) = a[p]; *(double2*)(smem + (buf) * STAGE + BK * LD + soff[p]) = b[p]; }
#define COMPUTE(buf) { const double* As = smem + (buf) * STAGE + (half * KH) * LD; const double* Bs = As + BK * LD; \
      _Pragma("unroll") for (int ks = 0; ks < KH; ks += 4) { double af[NTW], bf[NTW]; \
        _Pragma("unroll") for (int t = 0; t < NTW; ++t) { af[t] = As[(ks + kk) * LD + wy * WT + t * 16 + r16]; bf[t] = Bs[(ks + kk) * LD + wx * WT + t * 16 + r16]; } \
        _Pragma("unroll") for (int i = 0; i < NTW; ++i) _Pragma("unroll") for (int j = 0; j < NTW; ++j) { \
          if (MFMA) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0); else acc[i][j][0] += af[i] * bf[j]; } } }
    LOADSET(xa, xb);                // stage 0
    STORESET(xa, xb, 0);
    LOADSET(xa, xb);                // stage 1
    LOADSET(ya, yb);                // stage 2
    __syncthreads();
    // nk is odd or even; stages in pairs: s even uses X for s + 1, Y for s + 2
    for (int s = 0; s < nk; s += 2) {
      if (s + 1 < nk) { STORESET(xa, xb, 1); if (LOAD && s + 3 < nk) { LOADSET(xa, xb); } }
      COMPUTE(0);
      __syncthreads();
      if (s + 1 >= nk) break;
      if (s + 2 < nk) { STORESET(ya, yb, 0); if (LOAD && s + 4 < nk) { LOADSET(ya, yb); } }
      COMPUTE(1);
      __syncthreads();
    }
    if (half == 0) {
#pragma unroll
      for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = row0 + wy * WT + i * 16 + kk + 4 * r, col = col0 + wx * WT + j * 16 + r16;
            C[(size_t)row * N + col] = acc[i][j][r];
          }
    }
    __syncthreads();
  }
}
template <int TM, int PAD, bool LOAD, bool MFMA>
static int run2(const char* name, int N, double* A, double* B, double* C) {
  constexpr int LD = TM + PAD;
  const size_t lds = sizeof(double) * 2 * 2 * 32 * LD;
  auto kern = k2<TM, PAD, LOAD, MFMA>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int nb = N / TM, nt = nb * (nb + 1) / 2, grid = nt < 256 ? (nt + 7) / 8 * 8 : 256;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, N, A, B, C);
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, N, A, B, C);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1000.0 / reps, fl = 2.0 * TM * TM * (double)N * nt;
  printf("%-44s N=%d tiles=%d grid=%d  %.1f us  %.1f TFLOP/s\n", name, N, nt, grid, us, fl / us * 1e-6);
  return 0;
}

int main() {
  const int N = 2016;
  double *A, *B, *C;
  const size_t cap = 2048 * 2048;
  CK(hipMalloc(&A, sizeof(double) * cap)); CK(hipMalloc(&B, sizeof(double) * cap)); CK(hipMalloc(&C, sizeof(double) * cap));
  std::vector<double> h(cap);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1000) * 1e-3 - 0.5;
  CK(hipMemcpy(A, h.data(), sizeof(double) * cap, hipMemcpyHostToDevice)); CK(hipMemcpy(B, h.data(), sizeof(double) * cap, hipMemcpyHostToDevice));
  run<96, 8, true, true, 1>("TM=96 8 waves (k halves)", N, A, B, C);
  run<96, 8, false, true, 1>("TM=96 8 waves, no global loads", N, A, B, C);
  run<96, 8, true, false, 1>("TM=96 8 waves, no MFMA", N, A, B, C);
  run<96, 4, true, true, 1>("TM=96 4 waves", N, A, B, C);
  run<96, 4, false, true, 1>("TM=96 4 waves, no global loads", N, A, B, C);
  run<128, 8, true, true, 1>("TM=128 8 waves", 2048, A, B, C);
  run<128, 8, false, true, 1>("TM=128 8 waves, no global loads", 2048, A, B, C);
  run2<96, 16, true, true>("TM=96 two stages in flight", N, A, B, C);
  run2<96, 16, true, false>("TM=96 two stages in flight, no MFMA", N, A, B, C);
  run2<96, 0, false, true>("TM=96 unpadded LDS rows, no global loads", N, A, B, C);
  run2<96, 8, false, true>("TM=96 LDS rows + 8, no global loads", N, A, B, C);
  return 0;
}
