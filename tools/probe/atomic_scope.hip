// Float-atomic rate by memory scope (development aid).  Agent-scope atomics are carried out on the memory side
// of the fabric (~21 G line requests/s on MI355X); workgroup-scope atomics stay in the issuing XCD's L2.  They
// are only correct when every 128-byte line is touched by ONE XCD, so the table is sliced by line index and
// every XCD scans the whole batch for its slice.  The probe times both forms on the access pattern of
// hash_bwd_kernel (4 lanes share 16 adjacent bytes) and checks the sliced result against the agent-scope one.
// build: hipcc --offload-arch=gfx950 -O3 atomic_scope.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
__device__ __forceinline__ unsigned hsh(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ unsigned xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
  return x;
}
// element t (4 lanes per "corner pair"): entry e = pair base | x-offset, feature f
__device__ __forceinline__ void target(unsigned t, unsigned mask, unsigned& e, unsigned& f) {
  e = (hsh(t >> 2) & mask & ~1u) | ((t >> 1) & 1u);
  f = t & 1u;
}
// MODE 0: agent scope, every workgroup any line.  MODE 1: workgroup scope, every workgroup any line (WRONG sums:
// speed only).  MODE 2: sliced by line, workgroup scope.  MODE 3: sliced by line, agent scope.
template <int MODE>
__global__ void __launch_bounds__(512) k(float* tab, unsigned mask, unsigned n, unsigned* ctr, unsigned chunk) {
  if (MODE < 2) {
    for (unsigned t = blockIdx.x * 512 + threadIdx.x; t < n; t += gridDim.x * 512) {
      unsigned e, f;
      target(t, mask, e, f);
      if (MODE == 0) atomicAdd(tab + 2 * (size_t)e + f, 1.0f);
      else __hip_atomic_fetch_add(tab + 2 * (size_t)e + f, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    return;
  }
  const unsigned x = xcc_id();
  __shared__ unsigned c;
  const unsigned n_chunks = (n + chunk - 1) / chunk;
  for (;;) {
    if (threadIdx.x == 0) c = atomicAdd(ctr + x, 1u);
    __syncthreads();
    const unsigned my = c;
    __syncthreads();
    if (my >= n_chunks) break;
    const unsigned end = min(n, (my + 1) * chunk);
    for (unsigned t = my * chunk + threadIdx.x; t < end; t += 512) {
      unsigned e, f;
      target(t, mask, e, f);
      if (((e >> 4) & 7u) != x) continue;
      if (MODE == 3) atomicAdd(tab + 2 * (size_t)e + f, 1.0f);
      else __hip_atomic_fetch_add(tab + 2 * (size_t)e + f, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}
template <int MODE>
float run(float* tab, unsigned entries, unsigned n, unsigned* ctr, const char* name, int grid) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float total = 0.f;
  const int reps = 5;
  for (int i = 0; i < reps + 1; ++i) {
    (void)hipMemsetAsync(ctr, 0, 64, 0);
    (void)hipMemsetAsync(tab, 0, (size_t)entries * 8, 0);
    (void)hipEventRecord(e0);
    k<MODE><<<grid, 512>>>(tab, entries - 1, n, ctr, 8192);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (i) total += ms;
  }
  printf("%-44s %.3f ms per %u lane-atomics = %.1f G/s (%.1f G 16-byte requests/s)\n", name, total / reps, n, n * (double)reps / total * 1e-6,
         n / 4 * (double)reps / total * 1e-6);
  return total / reps;
}
int main(int argc, char** argv) {
  const unsigned log2e = argc > 1 ? atoi(argv[1]) : 19;
  const unsigned entries = 1u << log2e;
  const unsigned n = 4u * 2400000u;           // 200 k points x 12 hashed levels x 4 (dy, dz) instructions, 4 lanes each... one level's worth x 12
  float *a, *b; unsigned* ctr;
  if (hipMalloc(&a, (size_t)entries * 8) != hipSuccess || hipMalloc(&b, (size_t)entries * 8) != hipSuccess || hipMalloc(&ctr, 64) != hipSuccess) return 1;
  printf("table 2^%u entries (%.1f MB), %u lane-atomics\n", log2e, entries * 8e-6, n);
  run<0>(a, entries, n, ctr, "agent scope, unsliced", 2048);
  run<1>(b, entries, n, ctr, "workgroup scope, unsliced (wrong sums)", 2048);
  run<3>(b, entries, n, ctr, "agent scope, sliced by line per XCD", 2048);
  run<2>(b, entries, n, ctr, "workgroup scope, sliced by line per XCD", 2048);
  float* ha = (float*)malloc((size_t)entries * 8); float* hb = (float*)malloc((size_t)entries * 8);
  (void)hipMemcpy(ha, a, (size_t)entries * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(hb, b, (size_t)entries * 8, hipMemcpyDeviceToHost);
  double sa = 0, sb = 0; size_t bad = 0;
  for (size_t i = 0; i < (size_t)entries * 2; ++i) { sa += ha[i]; sb += hb[i]; bad += ha[i] != hb[i]; }
  printf("check: sum agent %.0f, sum sliced workgroup-scope %.0f (expected %u), differing entries %zu\n", sa, sb, n, bad);
  return bad != 0;
}
