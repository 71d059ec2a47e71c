// Micro-benchmark: what does a SIMD of gfx950 issue BESIDE a saturated stream of v_mfma_f64_16x16x4_f64?  Four wavefronts per
// SIMD (the occupancy of the C2 kernel), each running rounds of 6 independent MFMAs plus K other instructions of one kind:
//   kind 0: v_add_u32 (integer VALU)      kind 1: v_fma_f64 (fp64 VALU)      kind 2: ds_read_b64      kind 3: v_readlane_b32
//   kind 4: v_cndmask_b32                  kind 5: s_add_u32 (scalar ALU)
// If the other instructions were free (co-issued), ticks per round would stay at 4 x 6 x 64 = 1536 per SIMD until K is large;
// if they take issue cycles of the port the MFMAs use, every instruction adds its cycles.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_coissue.hip -o /tmp/mfma_coissue.exe && /tmp/mfma_coissue.exe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int KIND, int K>
__global__ __launch_bounds__(256) void k_mix(double* out, long long* ticks, int iters) {
  __shared__ double sm[512];
  sm[threadIdx.x] = threadIdx.x; sm[threadIdx.x + 256] = 1.0;
  __syncthreads();
  v4f64 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0;
  const double a = out[threadIdx.x & 63] + 1e-3, b = a + 1e-3;
  unsigned x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  double f0 = a, f1 = b, f2 = a + b, f3 = a - b;
  int s0 = blockIdx.x;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, a3, 0, 0, 0);
    a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, a4, 0, 0, 0);
    a5 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, b, a5, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < K; k += 4) {
      if (KIND == 0) {
        asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(x0));
      } else if (KIND == 1) {
        asm volatile("v_fma_f64 %0, %0, %4, %4\n\tv_fma_f64 %1, %1, %4, %4\n\tv_fma_f64 %2, %2, %4, %4\n\tv_fma_f64 %3, %3, %4, %4" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a));
      } else if (KIND == 2) {
        const unsigned ad = (threadIdx.x & 63) * 8;
        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:512\n\tds_read_b64 %2, %4 offset:1024\n\tds_read_b64 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(ad));
      } else if (KIND == 3) {
        int r0, r1, r2, r3;
        asm volatile("v_readlane_b32 %0, %4, 1\n\tv_readlane_b32 %1, %4, 2\n\tv_readlane_b32 %2, %4, 3\n\tv_readlane_b32 %3, %4, 4" : "=s"(r0), "=s"(r1), "=s"(r2), "=s"(r3) : "v"(x0));
        s0 += r0 + r1 + r2 + r3;
      } else if (KIND == 4) {
        asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(x0) : "vcc");
      } else {
        asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 3\n\ts_add_u32 %0, %0, 5\n\ts_add_u32 %0, %0, 7" : "+s"(s0) :: "scc");
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  const v4f64 s = a0 + a1 + a2 + a3 + a4 + a5;
  if (s[0] + s[1] + s[2] + s[3] + (double)(x0 + x1 + x2 + x3) + f0 + f1 + f2 + f3 + (double)s0 == 12345.678) out[threadIdx.x & 63] = s[0];
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int KIND, int K>
static void run(const char* name, int cus, double* out, long long* ticks) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f; long long t = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_mix<KIND, K>), dim3(cus * 4), dim3(256), 0, 0, out, ticks, iters);     // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) { best = ms; hipMemcpy(&t, ticks, sizeof t, hipMemcpyDeviceToHost); }
  }
  const double per_round = (double)t / iters;           // ticks per loop iteration of one wave = a round of the SIMD's 4 waves
  printf("  %-14s K = %3d per 6 MFMAs: %8.1f ticks per round (4 waves x 6 MFMAs = 1536 at 64 each) -> +%6.1f, %5.2f ticks per extra instruction and wave | %.1f TFLOP/s\n",
         name, K, per_round, per_round - 1536.0, K ? (per_round - 1536.0) / (4.0 * K) : 0.0, (double)cus * 16 * iters * 6 * 2048.0 / (best * 1e-3) / 1e12);
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double* out; long long* ticks;
  hipMalloc(&out, 64 * sizeof(double)); hipMemset(out, 0, 64 * sizeof(double));
  hipMalloc(&ticks, cus * 4 * sizeof(long long));
  printf("%s: %d CUs, 4 wavefronts per SIMD\n", p.gcnArchName, cus);
  run<0, 0>("(none)", cus, out, ticks);
  run<0, 8>("v_add_u32", cus, out, ticks); run<0, 24>("v_add_u32", cus, out, ticks); run<0, 48>("v_add_u32", cus, out, ticks); run<0, 96>("v_add_u32", cus, out, ticks);
  run<1, 8>("v_fma_f64", cus, out, ticks); run<1, 24>("v_fma_f64", cus, out, ticks); run<1, 48>("v_fma_f64", cus, out, ticks);
  run<2, 8>("ds_read_b64", cus, out, ticks); run<2, 24>("ds_read_b64", cus, out, ticks); run<2, 48>("ds_read_b64", cus, out, ticks);
  run<3, 8>("v_readlane", cus, out, ticks); run<3, 24>("v_readlane", cus, out, ticks); run<3, 48>("v_readlane", cus, out, ticks);
  run<4, 8>("v_cndmask", cus, out, ticks); run<4, 48>("v_cndmask", cus, out, ticks);
  run<5, 8>("s_add_u32", cus, out, ticks); run<5, 48>("s_add_u32", cus, out, ticks); run<5, 96>("s_add_u32", cus, out, ticks);
  return 0;
}
