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
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pg_gelu4(v2f& a, v2f& b) {   // two independent Horner chains, interleaved
  const v2f ca = {__builtin_amdgcn_fmed3f(a.x, -4.2f, 4.2f), __builtin_amdgcn_fmed3f(a.y, -4.2f, 4.2f)};
  const v2f cb = {__builtin_amdgcn_fmed3f(b.x, -4.2f, 4.2f), __builtin_amdgcn_fmed3f(b.y, -4.2f, 4.2f)};
  const v2f ua = ca * ca, ub = cb * cb;
  v2f qa = ua * -9.018102001e-10f + 7.941707090e-08f, qb = ub * -9.018102001e-10f + 7.941707090e-08f;
  constexpr float kC[6] = {-3.038026629e-06f, 6.689195681e-05f, -9.506666631e-04f, 9.298265605e-03f, -6.552827696e-02f, 3.984659427e-01f};
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    qa = qa * ua + kC[c];
    qb = qb * ub + kC[c];
  }
  a = a * (ca * qa + 0.5f);
  b = b * (cb * qb + 0.5f);
}

