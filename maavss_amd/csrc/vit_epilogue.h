// Shared by the K = 384 ViT GEMM kernels (vit_panel_gemm.hip, vit_ws_gemm.hip): the LDS-DMA copy and the polynomial GELU.
#pragma once
#include "common.h"

__device__ __forceinline__ void vit_glds16(const void* g, void* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// GELU for two accumulator values at a time.  gelu(v) = v * Phi(v), with Phi(v) - 1/2 = v * Q(v^2) on |v| <= 4.2 (Q: degree-7
// minimax fit constrained to Phi(+-4.2) = 1 / 0, so clamping the ARGUMENT is all that is needed outside) -- max |error| vs the
// exact-erf GELU 9.3e-5 over all v, an order of magnitude below the bf16 rounding of the stored value for |gelu| > 0.05.
// No transcendental, 11 full-rate instructions per PAIR (v_pk_mul/v_pk_fma): the erf form it replaces (rcp + exp + 9 FMAs per
// element) made fc1 VALU-bound at twice its MFMA time.
// Round 4: evaluated on SCALAR f32 instructions, not v_pk_mul_f32 / v_pk_fma_f32 -- beside the MFMAs of a SIMD's other waves a packed-f32
// instruction does not overlap with the matrix pipe at all, a plain v_fma_f32 / v_mul_f32 does (scripts/valu_probe, profiles/r4_valu_probe.txt:
// {24 MFMA + 64 x instruction} per wave at 3 waves per SIMD: v_pk_fma_f32 2455 ticks, v_fma_f32 1708, the MFMAs alone 1551); twice the
// instruction count, a third of the cost.  (The file is built with -fno-slp-vectorize so that hipcc does not re-pack them.)
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float pg_gelu1(float a) {
  const float c = __builtin_amdgcn_fmed3f(a, -4.2f, 4.2f);
  const float u = c * c;
  float q = u * -9.018102001e-10f + 7.941707090e-08f;
  constexpr float kC[6] = {-3.038026629e-06f, 6.689195681e-05f, -9.506666631e-04f, 9.298265605e-03f, -6.552827696e-02f, 3.984659427e-01f};
#pragma unroll
  for (int k = 0; k < 6; ++k) q = q * u + kC[k];
  return a * (c * q + 0.5f);
}
__device__ __forceinline__ void pg_gelu4(v2f& a, v2f& b) {   // four independent Horner chains (the compiler interleaves them)
  a.x = pg_gelu1(a.x); a.y = pg_gelu1(a.y); b.x = pg_gelu1(b.x); b.y = pg_gelu1(b.y);
}

// The same GELU on PACKED IEEE HALF (round 4, fc1 with IEEE-half storage).  Why: beside the MFMAs of the other waves of a SIMD,
// packed-f32 vector instructions do not overlap with the matrix pipe, packed-f16 ones do (scripts/valu_probe, profiles/r4_valu_probe.txt:
// {24 MFMA + 64 v_pk_fma_f32} per wave, 3 waves per SIMD: 2455 ticks = MORE than the sum of the two alone, 1551 + 582; the same with
// v_pk_fma_f16: 1668 = the MFMAs' 1551 + 8 %); fc1 543 -> 506 us per launch (same box).  The polynomial Q of pg_gelu4 is re-expanded in
// t = c^2 / 16 - 0.55 (t in [-0.55, 0.55]): in c^2 its coefficients span 9e-10 .. 0.4 (below the half range) and, scaled to O(1), its Horner
// chain cancels (intermediates up to 4.4 for a result of 0.13 .. 0.40: |error| up to 0.13 at |v| > 3 in half arithmetic); around the
// midpoint every coefficient is 0.15 .. 0.40 and the powers of t decay.  Error against the exact-erf GELU over v in [-6, 6], all of it
// rounding (input to half, seven half fmas, product): max 3.1e-3 (= 1.5 ulp of the half result at |v| 2 .. 3), N(0,1)-weighted rms 2.8e-4
// against 1.4e-4 for the f32 polynomial + one rounding (tests/tools/gelu_h2_error.py); what that does end to end is gated by the 1e-5 tests.
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pg_gelu_h2(v2f v) {
  const h2_t h = __builtin_convertvector(v, h2_t);
  const h2_t lim = {(_Float16)4.2f, (_Float16)4.2f};
  const h2_t c = __builtin_elementwise_max(__builtin_elementwise_min(h, lim), -lim);
  const h2_t t = (c * (_Float16)0.0625f) * c - (_Float16)0.55f;          // one v_pk_mul + one v_pk_fma
  constexpr float kT[8] = {-2.420778323e-01f, 4.003976983e-01f, -3.264899766e-01f, 2.595298127e-01f,
                           -2.277023313e-01f, 1.849680812e-01f, -1.484367893e-01f, 1.680398153e-01f};
  h2_t q = t * (_Float16)kT[0] + (_Float16)kT[1];
#pragma unroll
  for (int k = 2; k < 8; ++k) q = q * t + (_Float16)kT[k];
  const h2_t r = h * (c * q + (_Float16)0.5f);
  return __builtin_bit_cast(unsigned, r);
}
