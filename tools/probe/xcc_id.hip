#include <hip/hip_runtime.h>
__global__ void k(unsigned* out) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
  if (threadIdx.x == 0) out[blockIdx.x] = xcc;
}
int main() {
  unsigned* d; if (hipMalloc(&d, 4096 * 4) != hipSuccess) return 1;
  k<<<64, 64>>>(d);
  unsigned h[64]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; ++i) printf("%u ", h[i]);
  printf("\n");
  return 0;
}
