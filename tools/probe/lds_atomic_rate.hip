// LDS atomic throughput per CU on random addresses in a 64 KiB array (development aid): which ds_add flavour the
// binned hash-gradient reduce should use.  build: hipcc --offload-arch=gfx950 -O3 lds_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ unsigned hsh(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
  __shared__ float acc[16384];
  for (int i = threadIdx.x; i < 16384; i += 512) acc[i] = 0.f;
  __syncthreads();
  unsigned s = hsh(blockIdx.x * 512 + threadIdx.x);
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const unsigned a = (s >> 10) & 16383u;
    if (MODE == 0) atomicAdd(&acc[a], 1.0f);                                            // ds_add_f32
    if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(acc) + a, 1u);                 // ds_add_u32
    if (MODE == 2) atomicAdd(reinterpret_cast<unsigned long long*>(acc) + (a >> 1), 1ull);   // ds_add_u64
    if (MODE == 3) { float v = acc[a]; acc[a] = v + 1.0f; }                             // plain read + write (racy; rate only)
    if (MODE == 4) atomicAdd(&acc[(a & ~63u) | (threadIdx.x & 63u)], 1.0f);            // conflict-free banks
    if (MODE == 5) atomicAdd(&acc[a & ~63u], 1.0f);                                     // 64 lanes, one address
    if (MODE == 6) { const unsigned b = a & ~1u; atomicAdd(&acc[b], 1.0f); atomicAdd(&acc[b + 1], 2.0f); }   // the reduce's pair
  }
  __syncthreads();
  float t = 0.f;
  for (int i = threadIdx.x; i < 16384; i += 512) t += acc[i];
  if (t == -1.f) out[0] = t;
}
template <int MODE>
void run(float* out, const char* name) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 2000, grid = 512;
  k<MODE><<<grid, 512>>>(out, iters);
  (void)hipEventRecord(e0);
  k<MODE><<<grid, 512>>>(out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double ops = (double)grid * 512 * iters * (MODE == 6 ? 2 : 1);
  printf("%-36s %.3f ms  %.2f T lane-ops/s  = %.2f lanes/clk/CU at 2.4 GHz, 256 CUs\n", name, ms, ops / ms * 1e-9, ops / (ms * 1e-3) / 2.4e9 / 256);
}
int main() {
  float* out; if (hipMalloc(&out, 64) != hipSuccess) return 1;
  run<0>(out, "ds_add_f32 random");
  run<1>(out, "ds_add_u32 random");
  run<2>(out, "ds_add_u64 random");
  run<3>(out, "ds_read + ds_write random");
  run<4>(out, "ds_add_f32 conflict-free");
  run<5>(out, "ds_add_f32 one address per wave");
  run<6>(out, "ds_add_f32 adjacent pair");
  return 0;
}
