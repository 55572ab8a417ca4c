// Global atomic throughput on scattered addresses (development aid): fp32 pairs vs one packed fp16
// pair vs one 64-bit integer add per element.  build: hipcc --offload-arch=gfx950 -O3 atomic_rate.hip
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
__device__ __forceinline__ unsigned hsh(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int MODE>
__global__ void __launch_bounds__(512) k(float* t32, __half2* t16, unsigned long long* t64, unsigned mask, int n) {
  for (int i = blockIdx.x * 512 + threadIdx.x; i < n; i += gridDim.x * 512) {
    const unsigned e = hsh((unsigned)i) & mask;
    if (MODE == 0) { atomicAdd(t32 + 2 * (size_t)e, 1.0f); atomicAdd(t32 + 2 * (size_t)e + 1, 2.0f); }
    if (MODE == 1) atomicAdd(t32 + 2 * (size_t)e, 1.0f);
    if (MODE == 2) unsafeAtomicAdd(t16 + e, __floats2half2_rn(1.0f, 2.0f));
    if (MODE == 3) atomicAdd(t64 + e, 0x0000000100000002ull);
    // lanes in groups of 2 / 4 / 16 land in the same 128-byte line (adjacent half2 entries)
    if (MODE == 4) unsafeAtomicAdd(t16 + ((hsh((unsigned)i >> 1) & mask & ~1u) | (i & 1)), __floats2half2_rn(1.0f, 2.0f));
    if (MODE == 5) unsafeAtomicAdd(t16 + ((hsh((unsigned)i >> 2) & mask & ~3u) | (i & 3)), __floats2half2_rn(1.0f, 2.0f));
    if (MODE == 6) unsafeAtomicAdd(t16 + ((hsh((unsigned)i >> 4) & mask & ~15u) | (i & 15)), __floats2half2_rn(1.0f, 2.0f));
    if (MODE == 7) { const unsigned b = (hsh((unsigned)i >> 1) & mask & ~1u) | (i & 1); atomicAdd(t32 + 2 * (size_t)b, 1.0f); atomicAdd(t32 + 2 * (size_t)b + 1, 2.0f); }
  }
}
template <int MODE>
void run(float* t32, __half2* t16, unsigned long long* t64, unsigned mask, int n, const char* name) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<2048, 512>>>(t32, t16, t64, mask, n);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) k<MODE><<<2048, 512>>>(t32, t16, t64, mask, n);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %.3f ms per %d elements = %.1f G elements/s\n", name, ms / 5, n, n * 5.0 / ms * 1e-6);
}
int main() {
  const unsigned entries = 1u << 19;
  const int n = 25 * 1000 * 1000;
  float* t32; __half2* t16; unsigned long long* t64;
  if (hipMalloc(&t32, entries * 8) != hipSuccess || hipMalloc(&t16, entries * 4) != hipSuccess || hipMalloc(&t64, entries * 8) != hipSuccess) return 1;
  (void)hipMemset(t32, 0, entries * 8); (void)hipMemset(t16, 0, entries * 4); (void)hipMemset(t64, 0, entries * 8);
  run<0>(t32, t16, t64, entries - 1, n, "2 x fp32 atomic (pair)");
  run<1>(t32, t16, t64, entries - 1, n, "1 x fp32 atomic");
  run<2>(t32, t16, t64, entries - 1, n, "1 x packed fp16 pair");
  run<3>(t32, t16, t64, entries - 1, n, "1 x 64-bit integer add");
  run<4>(t32, t16, t64, entries - 1, n, "fp16 pair, 2 lanes per line");
  run<5>(t32, t16, t64, entries - 1, n, "fp16 pair, 4 lanes per line");
  run<6>(t32, t16, t64, entries - 1, n, "fp16 pair, 16 lanes per line");
  run<7>(t32, t16, t64, entries - 1, n, "2 x fp32, 2 lanes per line");
  return 0;
}
