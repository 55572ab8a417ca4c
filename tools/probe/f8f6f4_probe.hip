// v_mfma_scale_f32_32x32x64_f8f6f4 with A = e5m2 (cbsz 1), B = e4m3 (blgp 0) and unit scales, probed on hardware
// (development aid): (1) operand map -- assumed lane l, byte j (0..31) = A[row l&31][k = 32 (l>>5) + j],
// B[k = 32 (l>>5) + j][col l&31] -- checked with exact small integers; (2) that the eight dwords of an operand may
// be ANY consistent permutation of k between A and B (the wgrad kernel concatenates four ds_read_b64_tr_b8
// fragments); (3) issue rate against four v_mfma_f32_32x32x16_bf8_fp8 covering the same K.
// build: hipcc --offload-arch=gfx950 -O2 f8f6f4_probe.hip -o f8f6f4_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__global__ void mfma64_kernel(const i32x8* a, const i32x8* b, float* d) {
  const int l = threadIdx.x;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  // cbsz = 1: A is bf8 (e5m2); blgp = 0: B is fp8 (e4m3); scales: E8M0 127 = 2^0 in every byte
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], acc, 1, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int r = 0; r < 16; ++r) d[l * 16 + r] = acc[r];
}

template <int MODE>
__global__ void __launch_bounds__(256) rate_kernel(float* out, int iters) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  i32x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = 0x3c3c3c3c + threadIdx.x; b[e] = 0x38383838; }
  const long a1 = 0x3c3c3c3c3c3c3c3cL + threadIdx.x, b1 = 0x3838383838383838L;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (MODE == 0) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[t], 1, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      else {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf8_fp8(a1, b1, acc[t], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (s == -1.f) out[0] = s;
}

static unsigned char e4m3_of_int(int v) { static const unsigned char t[4] = {0x00, 0x38, 0x40, 0x44}; return (unsigned char)(t[abs(v)] | (v < 0 ? 0x80 : 0)); }
static unsigned char e5m2_of_int(int v) { static const unsigned char t[4] = {0x00, 0x3C, 0x40, 0x42}; return (unsigned char)(t[abs(v)] | (v < 0 ? 0x80 : 0)); }

int main() {
  static int A[32][64], B[64][32];
  srand(7);
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) A[i][k] = rand() % 7 - 3;
  for (int k = 0; k < 64; ++k) for (int j = 0; j < 32; ++j) B[k][j] = rand() % 7 - 3;
  i32x8 *da, *db; float* dd;
  hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dd, 64 * 16 * 4);
  for (int variant = 0; variant < 2; ++variant) {
    unsigned char ha[64][32], hb[64][32];
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 32; ++j) {
        // variant 0: natural k = 32 (l>>5) + j.  variant 1: the wgrad kernel's order -- dword pair g (bytes 8g..8g+7)
        // of lane half h holds the samples of old k-step g: k = 16 g + 8 h + (j & 7)
        const int h = l >> 5, k = variant == 0 ? 32 * h + j : 16 * (j >> 3) + 8 * h + (j & 7);
        ha[l][j] = e5m2_of_int(A[l & 31][k]);
        hb[l][j] = e4m3_of_int(B[k][l & 31]);
      }
    hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
    mfma64_kernel<<<1, 64>>>(da, db, dd);
    float hd[64 * 16];
    hipMemcpy(hd, dd, sizeof(hd), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        int ref = 0;
        for (int k = 0; k < 64; ++k) ref += A[row][k] * B[k][col];
        bad += hd[l * 16 + r] != (float)ref;
      }
    printf("32x32x64 f8f6f4, A e5m2 x B e4m3, unit scales, k order %s: %d / 1024 mismatches\n",
           variant == 0 ? "natural (k = 32 h + j)" : "four concatenated K=16 fragments (k = 16 (j>>3) + 8 h + (j&7))", bad);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float* out; hipMalloc(&out, 64);
  const int iters = 20000;
  for (int mode = 0; mode < 2; ++mode) {
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) rate_kernel<0><<<1024, 256>>>(out, iters); else rate_kernel<1><<<1024, 256>>>(out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double flop = 1024.0 * 4 * iters * 4 * (2.0 * 32 * 32 * 64);
    printf("%s: %.3f ms, %.0f TFLOP/s\n", mode == 0 ? "one 32x32x64 f8f6f4 per K=64" : "four 32x32x16 bf8_fp8 per K=64", ms, flop / ms * 1e-9);
  }
  return 0;
}
