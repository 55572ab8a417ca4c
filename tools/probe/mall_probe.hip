// Does the 256 MB Infinity Cache (MALL) serve a read that follows a write or a read of the same buffer?  (development aid)
// Per buffer size: read bandwidth (a) repeating the read, (b) after a plain-store fill of the buffer, (c) after a
// non-temporal-store fill, (d) after reading 2 GiB of other data (cold).
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/mall_probe.hip -o /tmp/mall_probe ; run: /tmp/mall_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) rd(const u32x4* __restrict__ p, size_t n_vec, unsigned* out) {
  u32x4 acc = {0, 0, 0, 0};
  const size_t stride = (size_t)gridDim.x * 256 * 8;
  for (size_t i = (size_t)blockIdx.x * 256 * 8 + threadIdx.x; i + 256 * 7 < n_vec; i += stride) {
    u32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + 256 * u);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
template <bool NT>
__global__ void __launch_bounds__(256) wr(u32x4* __restrict__ p, size_t n_vec, unsigned val) {
  const u32x4 v = {val, val, val, val};
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += stride) {
    if (NT) __builtin_nontemporal_store(v, p + i);
    else p[i] = v;
  }
}
static float timed(hipEvent_t e0, hipEvent_t e1) { float ms = 0.f; (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1); return ms; }
int main() {
  const size_t big = (size_t)2048 << 20;
  u32x4 *p, *other; unsigned* out;
  if (hipMalloc(&p, big) != hipSuccess || hipMalloc(&other, big) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
  (void)hipMemset(p, 1, big); (void)hipMemset(other, 2, big);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = 256 * 8;
  for (size_t mb : {32, 64, 128, 192, 256, 384, 512, 1024}) {
    const size_t bytes = mb << 20, nv = bytes / 16;
    float t[4];
    // (a) repeated read
    rd<<<grid, 256>>>(p, nv, out); rd<<<grid, 256>>>(p, nv, out);
    (void)hipEventRecord(e0); rd<<<grid, 256>>>(p, nv, out); (void)hipEventRecord(e1); t[0] = timed(e0, e1);
    // (b) after a plain-store fill, (c) after a non-temporal fill
    wr<false><<<grid, 256>>>(p, nv, 3u);
    (void)hipEventRecord(e0); rd<<<grid, 256>>>(p, nv, out); (void)hipEventRecord(e1); t[1] = timed(e0, e1);
    wr<true><<<grid, 256>>>(p, nv, 4u);
    (void)hipEventRecord(e0); rd<<<grid, 256>>>(p, nv, out); (void)hipEventRecord(e1); t[2] = timed(e0, e1);
    // (d) cold: 2 GiB of other data in between
    rd<<<grid, 256>>>(other, big / 16, out);
    (void)hipEventRecord(e0); rd<<<grid, 256>>>(p, nv, out); (void)hipEventRecord(e1); t[3] = timed(e0, e1);
    printf("%5zu MiB: read again %.2f TB/s | after plain fill %.2f | after nt fill %.2f | cold %.2f\n", mb, bytes / t[0] * 1e-9, bytes / t[1] * 1e-9,
           bytes / t[2] * 1e-9, bytes / t[3] * 1e-9);
  }
  return 0;
}
