// fp8 building blocks on gfx950, probed on hardware (development aid; results in profiles/r02_fp8_probe.txt):
//   1. v_cvt_scalef32_pk_{fp8,bf8}_bf16: scale direction, rounding, saturation, which half op_sel writes
//   2. ds_read_b64_tr_b8: which (source lane, byte) every destination byte comes from
//   3. v_mfma_f32_32x32x16_bf8_fp8: operand lane/byte -> (row, k) map, checked with exact integers
// build: hipcc --offload-arch=gfx950 -O2 fp8_probe.hip -o fp8_probe ; prints text to stdout
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

__global__ void cvt_kernel(const uint32_t* in, const float* scale, uint32_t* out, int n, int ovfl) {
  const int i = threadIdx.x;
  if (i >= n) return;
  if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");   // MODE.FP16_OVFL
  uint32_t a = 0xAAAAAAAAu, b = 0xAAAAAAAAu, c = 0xAAAAAAAAu, d = 0xAAAAAAAAu;
  asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(a) : "v"(in[i]), "v"(scale[i]));
  asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2 op_sel:[0,0,1]" : "+v"(b) : "v"(in[i]), "v"(scale[i]));
  asm volatile("v_cvt_scalef32_pk_bf8_bf16 %0, %1, %2" : "+v"(c) : "v"(in[i]), "v"(scale[i]));
  asm volatile("v_cvt_scalef32_pk_bf8_bf16 %0, %1, %2 op_sel:[0,0,1]" : "+v"(d) : "v"(in[i]), "v"(scale[i]));
  out[4 * i + 0] = a; out[4 * i + 1] = b; out[4 * i + 2] = c; out[4 * i + 3] = d;
}

// lane l supplies LDS address 8*perm(l); LDS byte a holds (hi ? a >> 8 : a & 255)
__global__ void tr_kernel(uint32_t* out, int variant) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[1024];
  const int l = threadIdx.x;
  for (int rep = 0; rep < 2; ++rep) {
    for (int a = l; a < 1024; a += 64) lds[a] = rep ? (unsigned char)(a >> 8) : (unsigned char)(a & 255);
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds + 8u * (unsigned)l;
    uint32_t v0, v1;
    uint64_t r;
    asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr) : "memory");
    v0 = (uint32_t)r; v1 = (uint32_t)(r >> 32);
    out[(rep * 64 + l) * 2 + 0] = v0;
    out[(rep * 64 + l) * 2 + 1] = v1;
    __syncthreads();
  }
}

// D = A(bf8) * B(fp8): lane l, byte j of each 64-bit operand given by the host
__global__ void mfma_kernel(const uint64_t* a, const uint64_t* b, float* d) {
  const int l = threadIdx.x;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  uint64_t av = a[l], bv = b[l];
  asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_bf8_fp8 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 3" : "+v"(acc) : "v"(av), "v"(bv));
  for (int r = 0; r < 16; ++r) d[l * 16 + r] = acc[r];
}

static unsigned char e4m3_of_int(int v) {   // |v| <= 3
  static const unsigned char t[4] = {0x00, 0x38, 0x40, 0x44};
  return (unsigned char)(t[v < 0 ? -v : v] | (v < 0 ? 0x80 : 0));
}
static unsigned char e5m2_of_int(int v) {
  static const unsigned char t[4] = {0x00, 0x3C, 0x40, 0x42};
  return (unsigned char)(t[v < 0 ? -v : v] | (v < 0 ? 0x80 : 0));
}

int main() {
  // ---- 1. conversions ----
  const float vals[][3] = {{1.0f, 2.0f, 1.0f}, {0.3f, -0.7f, 1.0f}, {1.0f, 2.0f, 2.0f}, {1.0f, 2.0f, 0.5f}, {448.f, 500.f, 1.0f},
                           {1e6f, -1e6f, 1.0f}, {0.001f, 0.003f, 1.0f}, {0.0009765625f, 0.001953125f, 1.0f}, {1.0625f, 1.1875f, 1.0f},
                           {1.125f, 1.375f, 1.0f}, {60000.f, 70000.f, 1.0f}, {1e-5f, 3e-5f, 1.0f}, {1e-5f, 3e-5f, 1.52587890625e-05f},
                           {0.0f, -0.0f, 1.0f}, {17.0f, 19.0f, 1.0f}, {1.0f, 2.0f, 65536.0f}, {__builtin_inff(), -__builtin_inff(), 1.0f}, {__builtin_nanf(""), 3.0f, 1.0f},
                           {3e38f, -3e38f, 5.9604644775390625e-08f}};
  const int n = sizeof(vals) / sizeof(vals[0]);
  uint32_t hin[64]; float hsc[64];
  for (int i = 0; i < n; ++i) { hin[i] = f2bf(vals[i][0]) | ((uint32_t)f2bf(vals[i][1]) << 16); hsc[i] = vals[i][2]; }
  uint32_t *din, *dout; float* dsc;
  hipMalloc(&din, 256); hipMalloc(&dsc, 256); hipMalloc(&dout, 64 * 16);
  hipMemcpy(din, hin, 4 * n, hipMemcpyHostToDevice); hipMemcpy(dsc, hsc, 4 * n, hipMemcpyHostToDevice);
  uint32_t hout[256];
  for (int ovfl = 0; ovfl < 2; ++ovfl) {
    cvt_kernel<<<1, 64>>>(din, dsc, dout, n, ovfl);
    hipMemcpy(hout, dout, 16 * n, hipMemcpyDeviceToHost);
    printf("== cvt_scalef32_pk_{fp8,bf8}_bf16, MODE.FP16_OVFL=%d: in (lo, hi) scale -> fp8 / fp8 op_sel / bf8 / bf8 op_sel (dst preset 0xAAAAAAAA)\n", ovfl);
    for (int i = 0; i < n; ++i)
      printf("(%g, %g) scale %g -> %08x %08x %08x %08x\n", vals[i][0], vals[i][1], vals[i][2], hout[4 * i], hout[4 * i + 1], hout[4 * i + 2], hout[4 * i + 3]);
  }

  // ---- 2. transposing 8-bit read ----
  uint32_t* dtr; hipMalloc(&dtr, 2 * 64 * 8);
  tr_kernel<<<1, 64>>>(dtr, 1);
  uint32_t htr[256];
  hipMemcpy(htr, dtr, 2 * 64 * 8, hipMemcpyDeviceToHost);
  printf("== ds_read_b64_tr_b8, lane l address 8*l: destination lane: 8 x (source lane, source byte)\n");
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 8; ++j) {
      const unsigned lo = (htr[(0 * 64 + l) * 2 + (j >> 2)] >> (8 * (j & 3))) & 255, hi = (htr[(1 * 64 + l) * 2 + (j >> 2)] >> (8 * (j & 3))) & 255;
      const unsigned addr = lo | (hi << 8);
      printf(" (%2u,%u)", addr >> 3, addr & 7);
    }
    printf("\n");
  }

  // ---- 3. MFMA bf8 x fp8 with the bf16-like operand map assumed ----
  int A[32][16], B[16][32];
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (int)((s >> 24) % 7) - 3; };
  for (int r = 0; r < 32; ++r) for (int k = 0; k < 16; ++k) A[r][k] = rnd();
  for (int k = 0; k < 16; ++k) for (int c = 0; c < 32; ++c) B[k][c] = rnd();
  uint64_t ha[64], hb[64];
  for (int l = 0; l < 64; ++l) {
    uint64_t av = 0, bv = 0;
    for (int j = 0; j < 8; ++j) {
      av |= (uint64_t)e5m2_of_int(A[l & 31][8 * (l >> 5) + j]) << (8 * j);
      bv |= (uint64_t)e4m3_of_int(B[8 * (l >> 5) + j][l & 31]) << (8 * j);
    }
    ha[l] = av; hb[l] = bv;
  }
  uint64_t *da, *db; float* dd;
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 64 * 64);
  hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
  mfma_kernel<<<1, 64>>>(da, db, dd);
  float hd[1024];
  hipMemcpy(hd, dd, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    int ref = 0;
    for (int k = 0; k < 16; ++k) ref += A[row][k] * B[k][col];
    if ((float)ref != hd[l * 16 + r]) ++bad;
  }
  printf("== v_mfma_f32_32x32x16_bf8_fp8 with A[row l&31][k 8(l>>5)+j] (e5m2), B[k 8(l>>5)+j][col l&31] (e4m3): %d / 1024 mismatches\n", bad);
  return hipDeviceSynchronize() == hipSuccess ? 0 : 1;
}
