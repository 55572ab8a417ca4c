// Achievable HBM read bandwidth on one MI355X with plain 16-byte loads (development aid).
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/hbm_read.hip -o /tmp/hbm_read ; run: /tmp/hbm_read
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) rd(const u32x4* __restrict__ p, size_t n_vec, unsigned* out) {
  u32x4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  for (size_t i = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i + 256 * (UNROLL - 1) < n_vec; i += stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + 256 * u) : p[i + 256 * u];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
template <int UNROLL, bool NT>
void run(const u32x4* p, size_t bytes, unsigned* out, int wgs_per_cu) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;
  for (int i = 0; i < 2; ++i) rd<UNROLL, NT><<<grid, 256>>>(p, bytes / 16, out);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) rd<UNROLL, NT><<<grid, 256>>>(p, bytes / 16, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("unroll %d nt %d wgs/cu %d: %.2f TB/s\n", UNROLL, (int)NT, wgs_per_cu, bytes * 10.0 / ms * 1e-9);
}
int main() {
  const size_t bytes = (size_t)2600 << 20;
  u32x4* p; unsigned* out;
  if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
  (void)hipMemset(p, 1, bytes);
  run<4, false>(p, bytes, out, 8); run<8, false>(p, bytes, out, 8); run<8, true>(p, bytes, out, 8);
  run<8, false>(p, bytes, out, 4); run<16, false>(p, bytes, out, 2); run<16, true>(p, bytes, out, 4);
  return 0;
}
