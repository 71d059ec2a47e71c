// Which XCD does a workgroup land on?  (HW_REG_XCC_ID, low 4 bits) for a 1-D grid and for a 2-D grid (x fastest).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  int x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.y * gridDim.x + blockIdx.x] = x;
}
int main() {
  int* d; hipMalloc(&d, sizeof(int) * 4096);
  int h[4096];
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, 0, d);
    hipMemcpy(h, d, sizeof(int) * 64, hipMemcpyDeviceToHost);
    printf("1-D grid of 64:");
    for (int i = 0; i < 64; ++i) printf(" %d", h[i] & 15);
    printf("\n");
  }
  hipLaunchKernelGGL(k, dim3(10, 9), dim3(256), 0, 0, d);
  hipMemcpy(h, d, sizeof(int) * 90, hipMemcpyDeviceToHost);
  printf("2-D grid (10, 9), linear order:");
  for (int i = 0; i < 90; ++i) printf(" %d", h[i] & 15);
  printf("\nraw value of block 0: 0x%x\n", h[0]);
  return 0;
}
