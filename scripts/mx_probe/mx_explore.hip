// Empirical discovery of the operand / scale layout of v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 x e4m3) on gfx950.
// The kernel takes raw register images; the host runs one-hot experiments.  Measurement tool, not product code.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void raw(const uint32_t* A, const uint32_t* B, const uint32_t* SA, const uint32_t* SB, float* D, int opsel) {
  const int lane = threadIdx.x;
  i32x8 a, b;
  for (int d = 0; d < 8; ++d) { a[d] = (int)A[lane * 8 + d]; b[d] = (int)B[lane * 8 + d]; }
  const int sa = (int)SA[lane], sb = (int)SB[lane];
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  f32x16 d;
  if (opsel == 0) d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  else if (opsel == 1) d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 1, sa, 1, sb);
  else if (opsel == 2) d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 2, sa, 2, sb);
  else d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 3, sa, 3, sb);
  for (int i = 0; i < 16; ++i) D[lane * 16 + i] = d[i];
}

static uint32_t hA[512], hB[512], hSA[64], hSB[64];
static float hD[1024];
static uint32_t *dA, *dB, *dSA, *dSB;
static float* dD;
static void run(int opsel) {
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipMemcpy(dSA, hSA, sizeof hSA, hipMemcpyHostToDevice); hipMemcpy(dSB, hSB, sizeof hSB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(raw, dim3(1), dim3(64), 0, 0, dA, dB, dSA, dSB, dD, opsel);
  hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
}
static void setbyte(uint32_t* img, int lane, int j, uint8_t v) {
  uint8_t* p = (uint8_t*)img;
  p[lane * 32 + j] = v;
}
static const uint8_t CODE[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50};   // e4m3 of 0..8

int main() {
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dSA, sizeof hSA); hipMalloc(&dSB, sizeof hSB); hipMalloc(&dD, sizeof hD);
  for (int i = 0; i < 64; ++i) hSA[i] = hSB[i] = 0x7f7f7f7fu;
  // ---- 1. D layout: A one-hot at (lane la, byte 0), B all ones -> which D (lane, reg) are non-zero; same for B
  printf("== D layout: A[lane la][byte 0] = 1, B = ones: non-zero D entries (lane: regs)\n");
  for (int la : {0, 1, 5, 31, 32, 37}) {
    memset(hA, 0, sizeof hA);
    memset(hB, 0x38, sizeof hB);
    setbyte(hA, la, 0, 0x38);
    run(0);
    printf("  la=%2d:", la);
    int shown = 0;
    for (int l = 0; l < 64 && shown < 6; ++l) {
      int any = 0;
      for (int r = 0; r < 16; ++r) if (hD[l * 16 + r] != 0.f) any = 1;
      if (any) { printf(" lane %d regs[", l); for (int r = 0; r < 16; ++r) if (hD[l * 16 + r] != 0.f) printf("%d:%g ", r, hD[l * 16 + r]); printf("]"); ++shown; }
    }
    printf("\n");
  }
  printf("== D layout: B[lane lb][byte 0] = 1, A = ones\n");
  for (int lb : {0, 1, 5, 31, 32, 37}) {
    memset(hB, 0, sizeof hB);
    memset(hA, 0x38, sizeof hA);
    setbyte(hB, lb, 0, 0x38);
    run(0);
    printf("  lb=%2d:", lb);
    int cnt = 0, firstl = -1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) if (hD[l * 16 + r] != 0.f) { ++cnt; if (firstl < 0) firstl = l; }
    printf(" %d non-zero entries, first lane %d; lane %d regs[", cnt, firstl, firstl);
    for (int r = 0; r < 16; ++r) if (hD[firstl * 16 + r] != 0.f) printf("%d:%g ", r, hD[firstl * 16 + r]);
    printf("]\n");
  }
  // ---- 2. k pairing: A one-hot at (lane 32 ha, byte ja) [row 0]; B column 0 lanes {0, 32} carry a position code
  printf("== k pairing: A (half ha, byte ja) pairs with B (half hb, byte jb)\n");
  int ok_identity = 1;
  for (int ha = 0; ha < 2; ++ha)
    for (int ja = 0; ja < 32; ++ja) {
      int dec[2];
      for (int pass = 0; pass < 2; ++pass) {
        memset(hA, 0, sizeof hA);
        memset(hB, 0, sizeof hB);
        setbyte(hA, 32 * ha, ja, 0x38);
        for (int hb = 0; hb < 2; ++hb)
          for (int jb = 0; jb < 32; ++jb) {
            const int pos = hb * 32 + jb;
            setbyte(hB, 32 * hb, jb, CODE[1 + (pass ? pos / 8 : pos % 8)]);
          }
        run(0);
        dec[pass] = (int)lrintf(hD[0]) - 1;     // D[row 0][col 0] = lane 0 reg 0 (checked above)
      }
      const int pos = dec[1] * 8 + dec[0];
      if (pos != ha * 32 + ja) ok_identity = 0;
      if (ja % 8 == 0 || pos != ha * 32 + ja) printf("  A(h%d, j%2d) <-> B(h%d, j%2d)%s\n", ha, ja, pos / 32, pos % 32, pos == ha * 32 + ja ? "" : "   <-- not identity");
    }
  printf("  pairing is %s\n", ok_identity ? "the identity: same (half, byte) of A and B meet" : "NOT the identity");
  // ---- 3. scales: A one-hot (lane la, byte ja) = 1, B ones; scale_a of lane l = 127 + (l % 7) - 3 in byte 0 (other bytes 0x7f)
  printf("== scale association (opsel 0): A[la][ja] = 1, B = ones, scale_a[l] = 2^((l %% 7) - 3): D tells whose scale applied\n");
  for (int la : {0, 1, 2, 3, 33, 34}) for (int ja : {0, 5, 16, 31}) {
    memset(hA, 0, sizeof hA);
    memset(hB, 0x38, sizeof hB);
    setbyte(hA, la, ja, 0x38);
    for (int l = 0; l < 64; ++l) { hSA[l] = 0x7f7f7f00u | (uint32_t)(127 + (l % 7) - 3); hSB[l] = 0x7f7f7f7fu; }
    run(0);
    float v = 0;
    for (int i = 0; i < 1024; ++i) if (hD[i] != 0.f) { v = hD[i]; break; }
    printf("  la=%2d ja=%2d: D = %g = 2^%g (own lane's scale would be 2^%d)\n", la, ja, v, log2f(v), (la % 7) - 3);
  }
  printf("== scale association for B: B[lb][jb] = 1, A = ones, scale_b[l] = 2^((l %% 7) - 3)\n");
  for (int lb : {0, 1, 2, 33, 34}) for (int jb : {0, 16, 31}) {
    memset(hB, 0, sizeof hB);
    memset(hA, 0x38, sizeof hA);
    setbyte(hB, lb, jb, 0x38);
    for (int l = 0; l < 64; ++l) { hSB[l] = 0x7f7f7f00u | (uint32_t)(127 + (l % 7) - 3); hSA[l] = 0x7f7f7f7fu; }
    run(0);
    float v = 0;
    for (int i = 0; i < 1024; ++i) if (hD[i] != 0.f) { v = hD[i]; break; }
    printf("  lb=%2d jb=%2d: D = %g = 2^%g (own lane's scale would be 2^%d)\n", lb, jb, v, log2f(v), (lb % 7) - 3);
  }
  // ---- 4. opsel: scale register bytes = {2^1, 2^2, 2^3, 2^4} in bytes 0..3; which one does opsel = s pick?
  printf("== opsel: scale_a bytes 0..3 = 2^1, 2^2, 2^3, 2^4 (all lanes), A = B = ones, K = 64 -> D = 64 * scale\n");
  for (int s = 0; s < 4; ++s) {
    memset(hA, 0x38, sizeof hA);
    memset(hB, 0x38, sizeof hB);
    for (int l = 0; l < 64; ++l) { hSA[l] = 0x83828180u; hSB[l] = 0x7f7f7f7fu; }
    run(s);
    printf("  opsel_a = opsel_b = %d: D[0] = %g -> scale_a applied = 2^%g\n", s, hD[0], log2f(hD[0] / 64.f));
  }
  for (int s = 0; s < 4; ++s) {
    for (int l = 0; l < 64; ++l) { hSB[l] = 0x83828180u; hSA[l] = 0x7f7f7f7fu; }
    run(s);
    printf("  opsel %d: scale_b applied = 2^%g\n", s, log2f(hD[0] / 64.f));
  }
  return 0;
}
