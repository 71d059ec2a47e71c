// Probe: operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950 (4 independent 4 x 4 x 4 products per instruction).
// For every pair (la, lb) of lanes: A = 1 in lane la only, B = 1 in lane lb only; the D lanes that come out non-zero tell which
// (block, i, k) x (block, k, j) the two lanes hold.  Checks the hypothesis
//     A[blk][i][k] : lane i + 4 blk + 16 k      B[blk][k][j] : lane j + 4 blk + 16 k      D[blk][i][j] : lane j + 4 blk + 16 i
// and prints every deviation.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_4x4_layout.hip -o tools/ubench/mfma_4x4_layout.exe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void probe(unsigned long long* mask) {
  const int la = blockIdx.x >> 6, lb = blockIdx.x & 63, lane = threadIdx.x;
  const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
  const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  const unsigned long long m = __ballot(d != 0.0);
  if (lane == 0) mask[blockIdx.x] = m;
}
int main() {
  unsigned long long* dm; static unsigned long long hm[4096];
  hipMalloc(&dm, sizeof hm);
  hipLaunchKernelGGL(probe, dim3(4096), dim3(64), 0, 0, dm);
  hipMemcpy(hm, dm, sizeof hm, hipMemcpyDeviceToHost);
  int bad = 0, hits = 0;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const int ia = la & 3, ba = (la >> 2) & 3, ka = la >> 4, jb = lb & 3, bb = (lb >> 2) & 3, kb = lb >> 4;
      const unsigned long long expect = (ba == bb && ka == kb) ? 1ull << (jb + 4 * ba + 16 * ia) : 0ull;
      if (hm[la * 64 + lb]) ++hits;
      if (hm[la * 64 + lb] != expect) { if (bad++ < 20) printf("la %2d lb %2d: mask %016llx expected %016llx\n", la, lb, hm[la * 64 + lb], expect); }
    }
  printf("pairs with a product: %d (expected 256); deviations from the hypothesis: %d\n", hits, bad);
  if (bad) {   // raw dump for the first lanes so that the true layout can be read off
    for (int la = 0; la < 64; la += 1)
      for (int lb = 0; lb < 64; ++lb) if (hm[la * 64 + lb]) printf("  A lane %2d x B lane %2d -> D mask %016llx\n", la, lb, hm[la * 64 + lb]);
  }
  return 0;
}
