// What bounds the one pass over inv(L22) (tail_solve.hip: ts_onepass_kernel, 4.4 TB/s in the kernel trace)?
// The production kernel's row mapping -- a 1024-thread workgroup holds a whole row in registers, rows interleaved over the workgroups, one
// workgroup-wide reduction per row -- rebuilt here with its pieces switchable:
//   MODE  0: loads only (every loaded value enters a sum; no LDS, no barrier)
//         1: the whole product x = W^T diag(d) W z (dot, wave sum, LDS exchange, barrier, accumulate)
//         2: as 1, a wavefront whose 64 NC columns lie above the diagonal skips its loads
//   D     rows in flight beyond the one being applied (ring of D + 1 register buffers; production: 1)
//   ORDER 0: a workgroup's rows longest first (production); 1: alternately from the long and the short end -- at every moment the chip
//            streams the same mix of long and short rows (no register cost)
// and a flat read of the same 4 K^2 bytes as the ceiling.   tri_stream.exe [K ...]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int NC>
struct Row {
  double2 w[NC / 2];
  __device__ __forceinline__ void load(const double* __restrict__ W, long long ld, int col0, int i) {
    const double* row = W + (long long)i * ld;
#pragma unroll
    for (int p = 0; p < NC / 2; ++p) {
      const int col = col0 + 128 * p;
      const double2 v = *reinterpret_cast<const double2*>(row + (col <= i ? col : 0));
      w[p].x = col <= i ? v.x : 0.0;
      w[p].y = col < i ? v.y : 0.0;
    }
  }
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int p = 0; p < NC / 2; ++p) w[p] = make_double2(0.0, 0.0);
  }
};

// MODE 3 = MODE 1 with tick stamps (s_memtime, 100 MHz): per workgroup, wavefront 0 accumulates the ticks it spends waiting for a row's data,
// forming its dot product and wave sum, at the barrier, and in the accumulate -- ticks[g * 8 + 0..3], rows in [4], whole kernel in [5]
template <int NC, int D, int MODE, int ORDER>
__global__ __launch_bounds__(1024) void tri_kernel(const double* __restrict__ W, long long ld, int K, const double* __restrict__ z,
                                                   const double* __restrict__ dinv, double* __restrict__ P, long long* __restrict__ ticks = nullptr) {
  extern __shared__ __attribute__((aligned(16))) double zs_all[];
  __shared__ double red[2][16];
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col0 = wave * (64 * NC) + 2 * lane;
  const int wcol = wave * (64 * NC);                 // first column of the wavefront's segment
  const int G = (int)gridDim.x, g = (int)blockIdx.x;
  const int count = g < K ? (K - g + G - 1) / G : 0;
  auto row_of = [&](int j) -> int {                  // r = K - 1 - i of the j-th row of this workgroup (clamped to its last)
    j = j < count ? j : count - 1;
    if (ORDER == 0) return g + j * G;
    const int h = j >> 1;
    return g + ((j & 1) ? (count - 1 - h) : h) * G;
  };
  Row<NC> buf[D + 1];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const int i = K - 1 - row_of(d);
    if (MODE == 2 && wcol > i) buf[d].zero(); else buf[d].load(W, ld, col0, i);
  }
  for (int c = tid; c < 1024 * NC; c += 1024) zs_all[c] = c < K ? z[c] : 0.0;
  double2 xa[NC / 2];
#pragma unroll
  for (int p = 0; p < NC / 2; ++p) xa[p] = make_double2(0.0, 0.0);
  __syncthreads();
  const double* zs = zs_all + col0;
  int j = 0, it = 0;
  long long tk0 = 0, tw = 0, td = 0, tb = 0, ta = 0;
  if (MODE == 3) tk0 = (long long)__builtin_amdgcn_s_memtime();
  while (j < count) {
#pragma unroll
    for (int s = 0; s <= D; ++s) {
      if (j < count) {
        {
          const int i = K - 1 - row_of(j + D);
          Row<NC>& nx = buf[(s + D) % (D + 1)];
          if (MODE == 2 && wcol > i) nx.zero(); else nx.load(W, ld, col0, i);
        }
        const Row<NC>& R = buf[s];
        const int i = K - 1 - row_of(j);
        if (MODE == 0) {
#pragma unroll
          for (int p = 0; p < NC / 2; ++p) { xa[p].x += R.w[p].x; xa[p].y += R.w[p].y; }
        } else {
          long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
          if (MODE == 3) {
            c0 = (long long)__builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D * (NC / 2)) : "memory");      // this row's data (the D newer rows stay in flight)
            c1 = (long long)__builtin_amdgcn_s_memtime();
          }
          double part = 0.0;
#pragma unroll
          for (int p = 0; p < NC / 2; ++p) {
            const double2 zz = *reinterpret_cast<const double2*>(zs + 128 * p);
            part += R.w[p].x * zz.x;
            part += R.w[p].y * zz.y;
          }
          part = wave_sum(part);
          if (lane == 0) red[it & 1][wave] = part;
          if (MODE == 3) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); c2 = (long long)__builtin_amdgcn_s_memtime(); }
          __syncthreads();
          if (MODE == 3) c3 = (long long)__builtin_amdgcn_s_memtime();
          const double* rr = red[it & 1];
          const double u = (((rr[0] + rr[1]) + (rr[2] + rr[3])) + ((rr[4] + rr[5]) + (rr[6] + rr[7]))) +
                           (((rr[8] + rr[9]) + (rr[10] + rr[11])) + ((rr[12] + rr[13]) + (rr[14] + rr[15])));
          const double vq = u * dinv[i];
#pragma unroll
          for (int p = 0; p < NC / 2; ++p) { xa[p].x += vq * R.w[p].x; xa[p].y += vq * R.w[p].y; }
          if (MODE == 3) {
            asm volatile("" ::"v"(xa[0].x));
            const long long c4 = (long long)__builtin_amdgcn_s_memtime();
            tw += c1 - c0; td += c2 - c1; tb += c3 - c2; ta += c4 - c3;
          }
          ++it;
        }
        ++j;
      }
    }
  }
#pragma unroll
  for (int p = 0; p < NC / 2; ++p) {
    const int col = col0 + 128 * p;
    if (col < K) *reinterpret_cast<double2*>(P + (size_t)g * K + col) = xa[p];
  }
  if (MODE == 3 && ticks && tid == 0) {
    long long* t = ticks + (size_t)g * 8;
    t[0] = tw; t[1] = td; t[2] = tb; t[3] = ta; t[4] = count; t[5] = (long long)__builtin_amdgcn_s_memtime() - tk0;
  }
}

__global__ void flat_kernel(const double2* __restrict__ src, size_t n2, double* __restrict__ out) {
  double a = 0.0, b = 0.0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n2; i += 4 * stride) {
    const double2 v0 = src[i], v1 = src[i + stride], v2 = src[i + 2 * stride], v3 = src[i + 3 * stride];
    a += (v0.x + v1.x) + (v2.x + v3.x);
    b += (v0.y + v1.y) + (v2.y + v3.y);
  }
  for (; i < n2; i += stride) { a += src[i].x; b += src[i].y; }
  if (a + b == 123.456) out[0] = a;
}

__global__ void fill_kernel(double* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned h = (unsigned)(i * 2654435761u) ^ seed;
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    p[i] = ((double)(h & 0xffff) / 65536.0 - 0.5) * 1e-2;
  }
}

__global__ void colsum_kernel(const double* __restrict__ P, int K, int G, double* __restrict__ x) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= K) return;
  double s = 0.0;
  for (int g = 0; g < G; ++g) s += P[(size_t)g * K + c];
  x[c] = s;
}

struct Ctx { double *W, *z, *dinv, *P, *x; long long* ticks; int K, G; };

template <int NC, int D, int MODE, int ORDER>
static void run(const Ctx& c, const std::vector<double>* ref, std::vector<double>* keep) {
  auto kern = tri_kernel<NC, D, MODE, ORDER>;
  const size_t lds = sizeof(double) * 1024 * (size_t)NC;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(c.G), dim3(1024), lds, 0, c.W, (long long)c.K, c.K, c.z, c.dinv, c.P, c.ticks);
  CK(hipDeviceSynchronize());
  const int reps = 20;
  CK(hipEventRecord(e0));
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(kern, dim3(c.G), dim3(1024), lds, 0, c.W, (long long)c.K, c.K, c.z, c.dinv, c.P, c.ticks);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, bytes = 4.0 * c.K * (double)c.K;
  double dev = -1.0;
  if (MODE != 0) {
    hipLaunchKernelGGL(colsum_kernel, dim3((c.K + 255) / 256), dim3(256), 0, 0, c.P, c.K, c.G, c.x);
    std::vector<double> x(c.K);
    CK(hipMemcpy(x.data(), c.x, sizeof(double) * c.K, hipMemcpyDeviceToHost));
    if (keep) *keep = x;
    if (ref) {
      double mx = 0.0, md = 0.0;
      for (int i = 0; i < c.K; ++i) { mx = fmax(mx, fabs((*ref)[i])); md = fmax(md, fabs(x[i] - (*ref)[i])); }
      dev = md / (mx > 0 ? mx : 1.0);
    }
  }
  printf("K %6d NC %2d MODE %d D %d ORDER %d | %7.1f us  %5.2f TB/s | vgpr %3d spill %d | dev %.1e\n", c.K, NC, MODE, D, ORDER, us, bytes / us * 1e-6,
         fa.numRegs, (int)fa.localSizeBytes, dev);
  if (MODE == 3) {
    std::vector<long long> t((size_t)c.G * 8);
    CK(hipMemcpy(t.data(), c.ticks, sizeof(long long) * t.size(), hipMemcpyDeviceToHost));
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int g = 0; g < c.G; ++g) for (int q = 0; q < 6; ++q) s[q] += (double)t[(size_t)g * 8 + q];
    // s_memtime ticks at 100 MHz: 10 ns each
    printf("      wavefront 0, mean over %d workgroups, us per kernel: wait-for-row %.1f  dot+wave-sum %.1f  barrier %.1f  accumulate %.1f  | rows %.1f  kernel %.1f\n", c.G,
           s[0] / c.G * 1e-2, s[1] / c.G * 1e-2, s[2] / c.G * 1e-2, s[3] / c.G * 1e-2, s[4] / c.G, s[5] / c.G * 1e-2);
  }
  fflush(stdout);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

template <int NC>
static void sweep(const Ctx& c) {
  std::vector<double> ref;
  run<NC, 1, 1, 0>(c, nullptr, &ref);      // production shape
  run<NC, 1, 3, 0>(c, &ref, nullptr);
  run<NC, 1, 3, 1>(c, &ref, nullptr);
  if constexpr (NC <= 12) run<NC, 2, 3, 1>(c, &ref, nullptr);
  run<NC, 1, 0, 0>(c, nullptr, nullptr);
  run<NC, 2, 0, 0>(c, nullptr, nullptr);
  run<NC, 1, 0, 1>(c, nullptr, nullptr);
  run<NC, 2, 0, 1>(c, nullptr, nullptr);
  run<NC, 1, 1, 1>(c, &ref, nullptr);
  run<NC, 1, 2, 0>(c, &ref, nullptr);
  run<NC, 1, 2, 1>(c, &ref, nullptr);
  if constexpr (NC <= 12) {
    run<NC, 2, 1, 0>(c, &ref, nullptr);
    run<NC, 2, 1, 1>(c, &ref, nullptr);
    run<NC, 2, 2, 1>(c, &ref, nullptr);
  }
  if constexpr (NC <= 8) {
    run<NC, 3, 0, 1>(c, nullptr, nullptr);
    run<NC, 3, 1, 0>(c, &ref, nullptr);
    run<NC, 3, 1, 1>(c, &ref, nullptr);
    run<NC, 3, 2, 1>(c, &ref, nullptr);
  }
}

int main(int argc, char** argv) {
  std::vector<int> Ks;
  for (int a = 1; a < argc; ++a) Ks.push_back(atoi(argv[a]));
  if (Ks.empty()) Ks = {7168, 9216, 15360};
  for (int K : Ks) {
    Ctx c;
    c.K = K; c.G = 256;
    CK(hipMalloc(&c.W, sizeof(double) * (size_t)K * K));
    CK(hipMalloc(&c.z, sizeof(double) * K)); CK(hipMalloc(&c.dinv, sizeof(double) * K)); CK(hipMalloc(&c.x, sizeof(double) * K));
    CK(hipMalloc(&c.P, sizeof(double) * (size_t)c.G * K));
    CK(hipMalloc(&c.ticks, sizeof(long long) * (size_t)c.G * 8));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, c.W, (size_t)K * K, 1u);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, c.z, (size_t)K, 2u);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, c.dinv, (size_t)K, 3u);
    CK(hipDeviceSynchronize());
    {   // the ceiling: the same 4 K^2 bytes as a flat stream (the lower half of the square's storage)
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      const size_t n2 = (size_t)K * K / 4;
      for (int grid : {1024, 2048, 4096}) {
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(flat_kernel, dim3(grid), dim3(256), 0, 0, (const double2*)c.W, n2, c.x);
        CK(hipEventRecord(e0));
        for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(flat_kernel, dim3(grid), dim3(256), 0, 0, (const double2*)c.W, n2, c.x);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("K %6d flat stream of 4 K^2 bytes, grid %4d x 256: %7.1f us  %5.2f TB/s\n", K, grid, ms * 1e3 / 20, 4.0 * K * (double)K / (ms * 1e3 / 20) * 1e-6);
      }
      CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
    const int nc = (K + 1023) / 1024;
    switch ((nc + 1) / 2) {
      case 3: sweep<6>(c); break;
      case 4: sweep<8>(c); break;
      case 5: sweep<10>(c); break;
      case 6: sweep<12>(c); break;
      case 7: sweep<14>(c); break;
      case 8: sweep<16>(c); break;
      default: printf("K %d: no instantiation\n", K); break;
    }
    CK(hipFree(c.W)); CK(hipFree(c.z)); CK(hipFree(c.dinv)); CK(hipFree(c.x)); CK(hipFree(c.P));
  }
  return 0;
}
