// Prototype (round 5): the Newton-Schulz step of the one-wavefront sign kernel on 32 x 32 blocks in two forms --
//   mode 0  the production form: 16 x 16 sub-tiles with v_mfma_f64_16x16x4_f64, 3 of 4 sub-tiles per symmetric product
//           (2 x 24 MFMAs of 64 cycles per step), Y kept in registers, its lower sub-tile transposed through LDS;
//   mode 1  4 x 4 granularity with v_mfma_f64_4x4x4_4b_f64 (four independent 4 x 4 x 4 products per instruction): a symmetric
//           product needs 36 of the 64 sub-blocks (10 accumulator registers of four blocks each: 40 block slots, 4 of them
//           duplicates), 80 MFMAs of 16 cycles = 1 280 cycles against 1 536; the operands with rotated block slots come from
//           LDS (the matrix is stored in full, leading dimension 34: every fragment load is base + immediate, conflict-free),
//           Y makes a round trip through LDS, and the step's combination rides in the second product: S' = S (alpha Y + beta I).
// Both run the same fixed schedule on random symmetric blocks; the results are compared with a host evaluation, and the kernel
// time per block and step is printed at the production occupancy (16 wavefronts per CU, 128 registers each).
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/sym4x4_proto.hip -o tools/ubench/sym4x4_proto.exe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int N = 32;
constexpr int NSTEP = 11;
__constant__ double c_mu[NSTEP];

__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
template <int CTRL>
__device__ __forceinline__ double sw_dpp(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sw_readlane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum(double v) {
  v += sw_dpp<0xB1>(v);
  v += sw_dpp<0x4E>(v);
  v += sw_dpp<0x141>(v);
  v += sw_dpp<0x140>(v);
  return (sw_readlane(v, 0) + sw_readlane(v, 16)) + (sw_readlane(v, 32) + sw_readlane(v, 48));
}

// ---------------------------------------------------------------------------------------------------------------------------
// mode 0: the production step (psd_sign_closed.h without the mirrored-read storage: full tile, LD = 33)
// ---------------------------------------------------------------------------------------------------------------------------
template <int LD>
__device__ __forceinline__ void step_big(double* __restrict__ T, int r16, int kk, double mu, bool stats, double& sa, double& sb, double& sg) {
  double f[8][2];
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int x = 0; x < 2; ++x) f[s][x] = T[(4 * s + kk) * LD + 16 * x + r16];
  wave_fence();
  v4f64 y[2][2];
  y[0][0] = y[0][1] = y[1][1] = v4f64{0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    y[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[s][0], f[s][0], y[0][0], 0, 0, 0);
    y[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[s][0], f[s][1], y[0][1], 0, 0, 0);
    y[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[s][1], f[s][1], y[1][1], 0, 0, 0);
  }
  double pa = 0, pb = 0;
  if (stats) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = i; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (i == j && kk + 4 * r == r16) pa += y[i][j][r];
          pb += (i == j ? 1.0 : 2.0) * (y[i][j][r] * y[i][j][r]);
        }
  }
  // lower sub-tile of Y through LDS (the tile is dead while the fragments are in registers)
#pragma unroll
  for (int r = 0; r < 4; ++r) T[r16 * 17 + kk + 4 * r] = y[0][1][r];
  wave_fence();
  v4f64 z[2][2];
  z[0][0] = z[0][1] = z[1][1] = v4f64{0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    z[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[s][0], y[0][0][s], z[0][0], 0, 0, 0);
    z[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[s][0], y[0][1][s], z[0][1], 0, 0, 0);
    z[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[s][1], y[0][1][s], z[1][1], 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) y[1][0][r] = T[(kk + 4 * r) * 17 + r16];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    z[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[4 + s][0], y[1][0][s], z[0][0], 0, 0, 0);
    z[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[4 + s][0], y[1][1][s], z[0][1], 0, 0, 0);
    z[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[4 + s][1], y[1][1][s], z[1][1], 0, 0, 0);
  }
  if (stats) {
    double pg = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = i; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double d = f[4 * i + r][j] - z[i][j][r];
          pg += (i == j ? 1.0 : 2.0) * (d * d);
        }
    sa = wave_sum(pa); sb = wave_sum(pb); sg = wave_sum(pg);
  }
  const double alpha = -0.5 * mu * mu * mu, beta = 1.5 * mu;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = i; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) z[i][j][r] = alpha * z[i][j][r] + beta * f[4 * i + r][j];
  wave_fence();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = i; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * i + kk + 4 * r, col = 16 * j + r16;
        if (i != j || col >= row) { T[row * LD + col] = z[i][j][r]; T[col * LD + row] = z[i][j][r]; }
      }
  wave_fence();
}

// ---------------------------------------------------------------------------------------------------------------------------
// mode 1: 4 x 4 x 4 blocks.  Lane = i + 4 b + 16 k (i, b, k in 0..3).  Operand A of block slot b: A_b[i][k]; operand B:
// B_b[k][j = i]; result D_b[i' = k][j = i] (probed: tools/ubench/mfma_4x4_layout.hip).
// F_s[c][K] (register, slot b) = M[16 c + 4 ((b + s) & 3) + i][4 K + k]: the A form of block (4 c + (b + s) & 3, K) of a symmetric M
// and, by symmetry, the B form of block (K, 4 c + (b + s) & 3).
//   mfma(F_0[c][K] of S, F_s[c'][K] of R) accumulates (S R) block (4 c + b, 4 c' + (b + s) & 3) in slot b.
// Accumulators: acc[0..2] = quadrant (0,0), s = 0, 1, 2; acc[3..5] = quadrant (1,1), s = 0, 1, 2; acc[6..9] = quadrant (0,1), s = 0..3.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int LDS1 = 34;
struct Lane4 {
  int baseN[4];    // (4 ((b + s) & 3) + i) * LD + k           fragment loads
  int baseD[4];    // (4 b + k) * LD + 4 ((b + s) & 3) + i     result stores, S in result form
  int b;
};
__device__ __forceinline__ Lane4 lane4_init(int lane) {
  Lane4 L;
  const int i = lane & 3, b = (lane >> 2) & 3, k = lane >> 4;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int bs = (b + s) & 3;
    L.baseN[s] = (4 * bs + i) * LDS1 + k;
    L.baseD[s] = (4 * b + k) * LDS1 + 4 * bs + i;
  }
  L.b = b;
  return L;
}

// One fragment load as a single ds_read_b64 (left to the compiler, pairs of loads become ds_read2_b64: half the LDS rate) whose
// completion is waited for by hand: the loads of k-block K + 1 are in flight while the ten MFMAs of k-block K issue.
template <int OFF>
__device__ __forceinline__ void lds_ld(double& r, unsigned addr) {
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
}
template <int CNT>
__device__ __forceinline__ void lds_wait(double (&b)[7]) {
  asm volatile("s_waitcnt lgkmcnt(%7)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]) : "n"(CNT));
}
template <bool SELF, int K>
__device__ __forceinline__ void symprod4_loads(const unsigned (&aN)[4], double (&b)[7]) {
  if (!SELF) { lds_ld<32 * K>(b[0], aN[0]); lds_ld<8 * (16 * LDS1 + 4 * K)>(b[3], aN[0]); }
  lds_ld<32 * K>(b[1], aN[1]);
  lds_ld<32 * K>(b[2], aN[2]);
  lds_ld<8 * (16 * LDS1 + 4 * K)>(b[4], aN[1]);
  lds_ld<8 * (16 * LDS1 + 4 * K)>(b[5], aN[2]);
  lds_ld<8 * (16 * LDS1 + 4 * K)>(b[6], aN[3]);
}
template <bool SELF, int K, int V>
__device__ __forceinline__ void symprod4_k(const double (&F0)[2][8], const unsigned (&aN)[4], double (&cur)[7], double (&nxt)[7], double (&acc)[10]) {
  constexpr int NL = SELF ? 5 : 7;
  if (!(V & 1)) {
    if (K < 7) symprod4_loads<SELF, (K < 7 ? K + 1 : 7)>(aN, nxt);
    if (K < 7) lds_wait<NL>(cur); else lds_wait<0>(cur);
  }
  const double b00 = SELF ? F0[0][K] : cur[0], b10 = SELF ? F0[1][K] : cur[3];
  acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(F0[0][K], b00, acc[0], 0, 0, 0);
  acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(F0[0][K], cur[1], acc[1], 0, 0, 0);
  acc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(F0[0][K], cur[2], acc[2], 0, 0, 0);
  acc[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(F0[1][K], b10, acc[3], 0, 0, 0);
  acc[4] = __builtin_amdgcn_mfma_f64_4x4x4f64(F0[1][K], cur[4], acc[4], 0, 0, 0);
  acc[5] = __builtin_amdgcn_mfma_f64_4x4x4f64(F0[1][K], cur[5], acc[5], 0, 0, 0);
  acc[6] = __builtin_amdgcn_mfma_f64_4x4x4f64(F0[0][K], b10, acc[6], 0, 0, 0);
  acc[7] = __builtin_amdgcn_mfma_f64_4x4x4f64(F0[0][K], cur[4], acc[7], 0, 0, 0);
  acc[8] = __builtin_amdgcn_mfma_f64_4x4x4f64(F0[0][K], cur[5], acc[8], 0, 0, 0);
  acc[9] = __builtin_amdgcn_mfma_f64_4x4x4f64(F0[0][K], cur[6], acc[9], 0, 0, 0);
}
template <bool SELF, int V>
__device__ __forceinline__ void symprod4(const double (&F0)[2][8], const double* __restrict__ T, const Lane4& L, double (&acc)[10]) {
  unsigned aN[4];
  const unsigned tb = (unsigned)(size_t)T;               // LDS byte address of the tile (the low 32 bits of the generic pointer)
#pragma unroll
  for (int s = 0; s < 4; ++s) aN[s] = tb + 8u * (unsigned)L.baseN[s];
  double b0[7], b1[7];
#pragma unroll
  for (int q = 0; q < 7; ++q) b0[q] = b1[q] = 0.0;
  if (V & 1) {
#pragma unroll
    for (int q = 0; q < 7; ++q) { b0[q] = F0[q & 1][q]; b1[q] = F0[(q + 1) & 1][q + 1]; }
  } else symprod4_loads<SELF, 0>(aN, b0);
  symprod4_k<SELF, 0, V>(F0, aN, b0, b1, acc);
  symprod4_k<SELF, 1, V>(F0, aN, b1, b0, acc);
  symprod4_k<SELF, 2, V>(F0, aN, b0, b1, acc);
  symprod4_k<SELF, 3, V>(F0, aN, b1, b0, acc);
  symprod4_k<SELF, 4, V>(F0, aN, b0, b1, acc);
  symprod4_k<SELF, 5, V>(F0, aN, b1, b0, acc);
  symprod4_k<SELF, 6, V>(F0, aN, b0, b1, acc);
  symprod4_k<SELF, 7, V>(F0, aN, b1, b0, acc);
}

// result registers -> the full symmetric matrix in LDS (position and mirror position; the duplicate slots of s = 2 stay silent)
__device__ __forceinline__ void store_sym4(double* __restrict__ T, const Lane4& L, int lane, const double (&acc)[10]) {
  const int i = lane & 3, b = L.b, k = lane >> 4;
#pragma unroll
  for (int q = 0; q < 2; ++q) {                       // diagonal quadrants
    const int o = q * (16 * LDS1 + 16);
    if (i >= k) {                                     // a diagonal block: its upper triangle decides (exactly symmetric iterate)
      T[L.baseD[0] + o] = acc[3 * q];
      T[(4 * b + i) * LDS1 + 4 * b + k + o] = acc[3 * q];
    }
    T[L.baseD[1] + o] = acc[3 * q + 1];
    T[(4 * ((b + 1) & 3) + i) * LDS1 + 4 * b + k + o] = acc[3 * q + 1];
    if (b < 2) {
      T[L.baseD[2] + o] = acc[3 * q + 2];
      T[(4 * ((b + 2) & 3) + i) * LDS1 + 4 * b + k + o] = acc[3 * q + 2];
    }
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    T[L.baseD[s] + 16] = acc[6 + s];
    T[(4 * ((b + s) & 3) + i) * LDS1 + 4 * b + k + 16 * LDS1] = acc[6 + s];
  }
}

template <int V>
__device__ __forceinline__ void step_small(double* __restrict__ T, const Lane4& L, int lane, double mu, bool stats, double& sa, double& sb, double& sg) {
  double F0[2][8];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int K = 0; K < 8; ++K) F0[c][K] = T[L.baseN[0] + 16 * c * LDS1 + 4 * K];
  double acc[10];
#pragma unroll
  for (int q = 0; q < 10; ++q) acc[q] = 0.0;
  symprod4<true, V>(F0, T, L, acc);
  const int i = lane & 3, k = lane >> 4;
  const bool diag = i == k;
  if (stats) {
    double pa = 0, pb = 0, pd = 0;
    const double w2 = L.b < 2 ? 2.0 : 0.0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const double y0 = acc[3 * q], y1 = acc[3 * q + 1], y2 = acc[3 * q + 2];
      if (diag) pa += y0;
      const double e0 = (diag ? 1.0 : 0.0) - y0;
      pb += y0 * y0 + 2.0 * (y1 * y1) + w2 * (y2 * y2);
      pd += e0 * e0 + 2.0 * (y1 * y1) + w2 * (y2 * y2);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) { pb += 2.0 * (acc[6 + s] * acc[6 + s]); pd += 2.0 * (acc[6 + s] * acc[6 + s]); }
    sa = wave_sum(pa); sb = wave_sum(pb); sg = wave_sum(pd);
  }
  const double alpha = -0.5 * mu * mu * mu, beta = 1.5 * mu;
#pragma unroll
  for (int q = 0; q < 10; ++q) acc[q] *= alpha;
  if (diag) { acc[0] += beta; acc[3] += beta; }
  wave_fence();                                      // every read of S from the tile has been issued and has landed
  if (!(V & 2)) store_sym4(T, L, lane, acc);
  else if (acc[0] + acc[5] + acc[9] == 1.2345) T[lane] = acc[1] + acc[2] + acc[3] + acc[4] + acc[6] + acc[7] + acc[8];
  wave_fence();
#pragma unroll
  for (int q = 0; q < 10; ++q) acc[q] = 0.0;
  symprod4<false, V>(F0, T, L, acc);
  wave_fence();
  if (!(V & 2)) store_sym4(T, L, lane, acc);
  else if (acc[0] + acc[5] + acc[9] == 1.2345) T[lane] = acc[1] + acc[2] + acc[3] + acc[4] + acc[6] + acc[7] + acc[8];
  wave_fence();
}

template <int MODE>
__global__ __launch_bounds__(256, 4) void ns_kernel(const double* __restrict__ in, double* __restrict__ out, double* __restrict__ stat, int nblocks, int reps) {
  constexpr int LD = MODE == 0 ? 33 : LDS1;
  __shared__ double tiles[4][N * LDS1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* T = tiles[wave];
  const int r16 = lane & 15, kk = lane >> 4;
  const Lane4 L = lane4_init(lane);
  const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  for (int rep = 0; rep < reps; ++rep)
    for (int blk = gw; blk < nblocks; blk += nw) {
      const double* src = in + (size_t)blk * N * N;
      double v[16], ss = 0;
#pragma unroll
      for (int u = 0; u < 16; ++u) { v[u] = src[lane + 64 * u]; ss += v[u] * v[u]; }
      const double scale = 1.0 / sqrt(wave_sum(ss));
#pragma unroll
      for (int u = 0; u < 16; ++u) { const int e = lane + 64 * u; T[(e >> 5) * LD + (e & 31)] = v[u] * scale; }
      wave_fence();
      double sa = 0, sb = 0, sg = 0, acc_stat = 0;
#pragma unroll 1
      for (int st = 0; st < NSTEP; ++st) {
        const bool stats = st == 0 || st >= 5;
        if (MODE == 0) step_big<LD>(T, r16, kk, c_mu[st], stats, sa, sb, sg);
        else step_small<(MODE > 0 ? MODE - 1 : 0)>(T, L, lane, c_mu[st], stats, sa, sb, sg);
        acc_stat += sa + sb + sg;
      }
      double* dst = out + (size_t)blk * N * N;
#pragma unroll
      for (int u = 0; u < 16; ++u) { const int e = lane + 64 * u; dst[e] = T[(e >> 5) * LD + (e & 31)]; }
      if (lane == 0) stat[blk] = acc_stat;
      wave_fence();
    }
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 8, reps = argc > 2 ? atoi(argv[2]) : 4;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, nblocks = cus * 16 * rounds;
  std::vector<double> h((size_t)nblocks * N * N), mu(NSTEP);
  std::mt19937_64 rng(1);
  std::normal_distribution<double> nd;
  for (int b = 0; b < nblocks; ++b) {
    double* M = &h[(size_t)b * N * N];
    for (int i = 0; i < N; ++i)
      for (int j = 0; j <= i; ++j) { const double x = nd(rng); M[i * N + j] = x; M[j * N + i] = x; }
  }
  const double sched[NSTEP] = {1.53 / 0.45, 1.53, 1.53, 1.53, 1.309, 1.0845, 1.0, 1.0, 1.0, 1.0, 1.0};   // a typical C2 block: lifts, probes, plain steps
  hipMemcpyToSymbol(HIP_SYMBOL(c_mu), sched, sizeof sched);
  double *din, *dout, *dstat;
  hipMalloc(&din, h.size() * 8); hipMalloc(&dout, h.size() * 8); hipMalloc(&dstat, nblocks * 8);
  hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<double> res[2];
  printf("%s: %d CUs, %d blocks of 32 x 32 (%d per wavefront slot), %d steps, %d repetitions per launch\n", p.gcnArchName, cus, nblocks, rounds, NSTEP, reps);
  for (int mode = 0; mode < 5; ++mode) {
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(ns_kernel<0>, dim3(cus * 4), dim3(256), 0, 0, din, dout, dstat, nblocks, reps);
      else if (mode == 1) hipLaunchKernelGGL(ns_kernel<1>, dim3(cus * 4), dim3(256), 0, 0, din, dout, dstat, nblocks, reps);
      else if (mode == 2) hipLaunchKernelGGL(ns_kernel<2>, dim3(cus * 4), dim3(256), 0, 0, din, dout, dstat, nblocks, reps);
      else if (mode == 3) hipLaunchKernelGGL(ns_kernel<3>, dim3(cus * 4), dim3(256), 0, 0, din, dout, dstat, nblocks, reps);
      else hipLaunchKernelGGL(ns_kernel<4>, dim3(cus * 4), dim3(256), 0, 0, din, dout, dstat, nblocks, reps);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (it > 0 && ms < best) best = ms;
    }
    if (hipGetLastError() != hipSuccess) { printf("mode %d: launch failed\n", mode); return 1; }
    if (mode < 2) { res[mode].resize(h.size()); hipMemcpy(res[mode].data(), dout, h.size() * 8, hipMemcpyDeviceToHost); }
    const double per_step_ns = best * 1e6 / ((double)nblocks * reps * NSTEP);
    printf("mode %d (%s): %.3f ms per launch, %.2f ns per block and step on the whole chip = %.0f ns of a wavefront slot per step\n", mode,
           mode == 0 ? "16x16x4 tiles" : mode == 1 ? "4x4x4 blocks" : mode == 2 ? "4x4x4, no B loads (wrong results)" : mode == 3 ? "4x4x4, no stores (wrong results)" : "4x4x4, neither", best, per_step_ns, per_step_ns * cus * 16);
  }
  // host evaluation of the schedule on the first blocks
  double worst[2] = {0, 0}, asym[2] = {0, 0};
  for (int b = 0; b < 4; ++b) {
    std::vector<double> S(h.begin() + (size_t)b * N * N, h.begin() + (size_t)(b + 1) * N * N), Y(N * N), Z(N * N);
    double ss = 0;
    for (double x : S) ss += x * x;
    for (double& x : S) x /= std::sqrt(ss);
    for (int st = 0; st < NSTEP; ++st) {
      for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) { double a = 0; for (int k = 0; k < N; ++k) a += S[i * N + k] * S[k * N + j]; Y[i * N + j] = a; }
      for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) { double a = 0; for (int k = 0; k < N; ++k) a += S[i * N + k] * Y[k * N + j]; Z[i * N + j] = a; }
      const double m = sched[st], al = -0.5 * m * m * m, be = 1.5 * m;
      for (int e = 0; e < N * N; ++e) S[e] = al * Z[e] + be * S[e];
    }
    for (int mode = 0; mode < 2; ++mode)
      for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
          const double g = res[mode][(size_t)b * N * N + i * N + j];
          worst[mode] = std::fmax(worst[mode], std::fabs(g - S[i * N + j]));
          asym[mode] = std::fmax(asym[mode], std::fabs(g - res[mode][(size_t)b * N * N + j * N + i]));
        }
  }
  printf("max |device - host| over 4 blocks: mode 0 %.2e, mode 1 %.2e; asymmetry %.1e / %.1e\n", worst[0], worst[1], asym[0], asym[1]);
  return 0;
}
