// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands (gfx950): checks, with exact small-integer data and non-uniform
// e8m0 block scales, the operand layout the fp8 attention kernel (maavss_amd/csrc/vit_attn_mx.hip) relies on:
//   lane l (r = l & 31, h = l >> 5) holds A[row r][k = 32 h + j] / B[k = 32 h + j][col r] in byte j (j = 0..31) of its 8 VGPRs,
//   its scale operand (byte `opsel` of the scale VGPR, e8m0) multiplies exactly those 32 values,
//   C/D: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
// Build: hipcc --offload-arch=gfx950 -O2 mx_probe.hip -o mx_probe.bin ; run on the GPU box.  Measurement tool, not product code.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void probe(const uint8_t* A, const uint8_t* B, const uint8_t* sa, const uint8_t* sb, float* D, int opsel) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  i32x8 a, b;
  for (int d = 0; d < 8; ++d) {
    uint32_t wa = 0, wb = 0;
    for (int e = 0; e < 4; ++e) {
      const int k = 32 * h + 4 * d + e;
      wa |= (uint32_t)A[r * 64 + k] << (8 * e);
      wb |= (uint32_t)B[k * 32 + r] << (8 * e);
    }
    a[d] = (int)wa;
    b[d] = (int)wb;
  }
  // scale VGPRs: byte `opsel` carries this lane's e8m0 scale, the other bytes garbage that must be ignored
  const uint32_t va = 0x11223344u, vb = 0x55667788u;
  const int sh = 8 * opsel;
  const int scale_a = (int)((va & ~(0xffu << sh)) | ((uint32_t)sa[r * 2 + h] << sh));
  const int scale_b = (int)((vb & ~(0xffu << sh)) | ((uint32_t)sb[h * 32 + r] << sh));
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  f32x16 d;
  if (opsel == 0) d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
  else if (opsel == 1) d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 1, scale_a, 1, scale_b);
  else if (opsel == 2) d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 2, scale_a, 2, scale_b);
  else d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 3, scale_a, 3, scale_b);
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    D[row * 32 + r] = d[i];
  }
}

// timing: a dependent chain of N MFMAs per wave, many waves -> cycles per instruction
__global__ void chain_scaled(float* out, int n) {
  i32x8 a, b;
  for (int d = 0; d < 8; ++d) { a[d] = 0x38383838; b[d] = 0x38383838 + threadIdx.x * 0; }
  f32x16 c0, c1;
  for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
  for (int i = 0; i < n; ++i) {
    c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[3];
}
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__global__ void chain_f16(float* out, int n) {
  f16x8 a, b;
  for (int d = 0; d < 8; ++d) { a[d] = (_Float16)1.f; b[d] = (_Float16)1.f; }
  f32x16 c0, c1;
  for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
  for (int i = 0; i < n; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[3];
}

static float e4m3_decode(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f;
  if (e == 0) f = ldexpf((float)m, -9);
  else if (e == 15 && m == 7) f = NAN;
  else f = ldexpf(1.f + m / 8.f, e - 7);
  return s ? -f : f;
}

int main() {
  uint8_t hA[32 * 64], hB[64 * 32], hsa[64], hsb[64];
  float want[32 * 32], got[32 * 32];
  uint8_t ints[9];   // e4m3 codes of -4..4 (exact)
  for (int v = -4; v <= 4; ++v)
    for (int c = 0; c < 256; ++c)
      if (e4m3_decode((uint8_t)c) == (float)v && !(v == 0 && c != 0)) ints[v + 4] = (uint8_t)c;
  srand(7);
  for (int i = 0; i < 32 * 64; ++i) hA[i] = ints[rand() % 9];
  for (int i = 0; i < 64 * 32; ++i) hB[i] = ints[rand() % 9];
  for (int i = 0; i < 64; ++i) { hsa[i] = 127 + (rand() % 7) - 3; hsb[i] = 127 + (rand() % 5) - 2; }
  for (int r = 0; r < 32; ++r)
    for (int c = 0; c < 32; ++c) {
      double acc = 0;
      for (int k = 0; k < 64; ++k)
        acc += (double)e4m3_decode(hA[r * 64 + k]) * ldexp(1.0, hsa[r * 2 + k / 32] - 127) * e4m3_decode(hB[k * 32 + c]) * ldexp(1.0, hsb[(k / 32) * 32 + c] - 127);
      want[r * 32 + c] = (float)acc;
    }
  uint8_t *dA, *dB, *dsa, *dsb;
  float* dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dD, sizeof got);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipMemcpy(dsa, hsa, 64, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 64, hipMemcpyHostToDevice);
  int fails = 0;
  for (int opsel = 0; opsel < 4; ++opsel) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD, opsel);
    hipMemcpy(got, dD, sizeof got, hipMemcpyDeviceToHost);
    int bad = 0;
    double worst = 0;
    for (int i = 0; i < 1024; ++i) {
      const double e = fabs((double)got[i] - want[i]);
      if (e > 1e-3 * (1 + fabs(want[i]))) ++bad;
      if (e > worst) worst = e;
    }
    printf("opsel %d: %d of 1024 outputs differ from the layout hypothesis (worst |err| %.3g; sample got %.3f want %.3f)\n", opsel, bad, worst, got[37], want[37]);
    fails += bad;
  }
  // rate: 1024 blocks x 256 threads, 2 x n MFMAs per wave
  float* dout;
  hipMalloc(&dout, 1024 * 256 * 4);
  const int n = 4096;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(chain_scaled, dim3(1024), dim3(256), 0, 0, dout, n);
      else hipLaunchKernelGGL(chain_f16, dim3(1024), dim3(256), 0, 0, dout, n);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flop = 1024.0 * 4 * 2 * n * 2.0 * 32 * 32 * (which == 0 ? 64 : 16);
      if (rep) printf("%s: %.3f ms, %.0f TFLOP/s (all-ones operands)\n", which == 0 ? "v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3)" : "v_mfma_f32_32x32x16_f16", ms, flop / ms / 1e9);
    }
  }
  printf(fails ? "LAYOUT HYPOTHESIS FAILED\n" : "layout hypothesis holds\n");
  return fails ? 1 : 0;
}
