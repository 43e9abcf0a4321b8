// Shared device/host helpers for libmaavss_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define MAAVSS_OK 0
#define MAAVSS_ERR_ARG 1
#define MAAVSS_ERR_LAUNCH 2

void maavss_set_error(const char* fmt, ...);
int maavss_deterministic_flag(void);
float* maavss_deterministic_ws(int64_t* floats, void* stream);   // api_core.hip: scratch for deterministic split-K partials (null if none was provided for `stream`)

#define MAAVSS_CHECK_ARG(cond, ...)            \
  do {                                         \
    if (!(cond)) {                             \
      maavss_set_error(__VA_ARGS__);           \
      return MAAVSS_ERR_ARG;                   \
    }                                          \
  } while (0)

#define MAAVSS_LAUNCH_CHECK(name)                                              \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess) {                                                   \
      maavss_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return MAAVSS_ERR_LAUNCH;                                                \
    }                                                                          \
  } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ bf16_t f2bf(float f) {
  // round-to-nearest-even; NaN stays NaN (compiler emits v_cvt_pk_bf16_f32)
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((unsigned)b) << 16); }
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  // one v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN preserved) instead of two converts plus a shift/or
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// 16-bit storage formats of the ViT path (include/maavss.h `dtype`): 0 = bf16, 2 = IEEE half (same MFMA rate, 3 more
// mantissa bits, range +-65504 -- the LayerNorm-ed / bounded activations of the extractor fit).  Numbering = MODE_* of mma.h.
template <int MODE>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  if constexpr (MODE == 2) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2_t));   // round-to-nearest-even per element
  } else {
    return pack_bf2(lo, hi);
  }
}
template <int MODE>
__device__ __forceinline__ unsigned short cvt16(float f) {
  if constexpr (MODE == 2) {
    const _Float16 hv = (_Float16)f;
    return __builtin_bit_cast(unsigned short, hv);
  } else {
    return f2bf(f);
  }
}
template <int MODE>
__device__ __forceinline__ float up16(unsigned short b) {
  if constexpr (MODE == 2) return (float)__builtin_bit_cast(_Float16, b);
  else return bf2f(b);
}

// Two 4 x 16-bit halves (e.g. two ds_read_b64_tr_b16 results) -> one 8 x 16-bit MFMA fragment, as a pure
// register-pair concatenation (an element-wise vector initialiser makes hipcc emit per-element pack code).
__device__ __forceinline__ bf16x8 concat4(bf16x4 lo, bf16x4 hi) {
  const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
  const uint4 u = make_uint4(a.x, a.y, b.x, b.y);
  return __builtin_bit_cast(bf16x8, u);
}

// Wave-wide reductions on the VALU (DPP), result uniform across the wave.  The __shfl_xor form goes through the LDS
// crossbar: six dependent ds_bpermute round trips (~500 cycles per reduction, stamped in the LayerNorm prologue).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float x, float old) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f32<0xB1, 0xf>(v, 0.f);    // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E, 0xf>(v, 0.f);    // quad_perm [2,3,0,1]
  v += dpp_f32<0x141, 0xf>(v, 0.f);   // row_half_mirror
  v += dpp_f32<0x140, 0xf>(v, 0.f);   // row_mirror: every lane of a 16-lane row holds the row sum
  v += dpp_f32<0x142, 0xa>(v, 0.f);   // row_bcast:15 into rows 1, 3
  v += dpp_f32<0x143, 0xc>(v, 0.f);   // row_bcast:31 into rows 2, 3: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f32<0xB1, 0xf>(v, v));
  v = fmaxf(v, dpp_f32<0x4E, 0xf>(v, v));
  v = fmaxf(v, dpp_f32<0x141, 0xf>(v, v));
  v = fmaxf(v, dpp_f32<0x140, 0xf>(v, v));
  v = fmaxf(v, dpp_f32<0x142, 0xa>(v, v));   // masked-off rows keep their own value (old = v)
  v = fmaxf(v, dpp_f32<0x143, 0xc>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// sum / max over the four lanes {l, l^16, l^32, l^48} (the four 16-lane rows of a wave) through the gfx950 lane-swap
// instructions -- one VALU op per exchange instead of a ds_bpermute round trip through the LDS crossbar.
// v_permlane16_swap a, b exchanges IN PLACE the odd 16-lane rows of a with the even rows of b (v_permlane32_swap: the
// upper half of a with the lower half of b): starting from two copies of v, a = {v0, v0, v2, v2} and b = {v1, v1, v3, v3}
// (rows).  Written as inline asm: with this hipcc the second result of __builtin_amdgcn_permlane16_swap reads back as the
// first one (v + v instead of v_even + v_odd in the emitted code), checked on the GPU.
__device__ __forceinline__ void lane_swap16(float v, float& a, float& b) {
  a = v;
  b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void lane_swap32(float v, float& a, float& b) {
  a = v;
  b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float rows4_sum(float v) {
  float a, b;
  lane_swap16(v, a, b);
  lane_swap32(a + b, a, b);
  return a + b;
}
// sum over the 16 lanes of a lane's DPP row
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f32<0xB1, 0xf>(v, 0.f);
  v += dpp_f32<0x4E, 0xf>(v, 0.f);
  v += dpp_f32<0x141, 0xf>(v, 0.f);
  v += dpp_f32<0x140, 0xf>(v, 0.f);
  return v;
}
// sum over the 32 consecutive lanes [0, 32) / [32, 64) a lane belongs to (DPP inside the 16-lane rows, one lane swap across)
__device__ __forceinline__ float half_wave_sum(float v) {
  v += dpp_f32<0xB1, 0xf>(v, 0.f);
  v += dpp_f32<0x4E, 0xf>(v, 0.f);
  v += dpp_f32<0x141, 0xf>(v, 0.f);
  v += dpp_f32<0x140, 0xf>(v, 0.f);
  float a, b;
  lane_swap16(v, a, b);
  return a + b;
}
__device__ __forceinline__ float rows4_max(float v) {
  float a, b;
  lane_swap16(v, a, b);
  lane_swap32(fmaxf(a, b), a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Philox4x32-10 counter RNG (Salmon et al. 2011) -> 4 x N(0,1) via Box-Muller.
__device__ __forceinline__ uint64_t mul_wide_u32(uint32_t a, uint32_t k) {
  uint64_t p;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "v"(a), "s"(k) : "vcc");
  return p;
}
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a v_mul_hi_u32 / v_mul_lo_u32 pair: integer multiplies are quarter rate
    const uint64_t p0 = mul_wide_u32(c[0], 0xD2511F53u), p1 = mul_wide_u32(c[2], 0xCD9E8D57u);
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t idx, float out[4]) {
  uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), 0x5A17u, 0u};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    float u1 = ((float)(c[2 * p] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    float u2 = ((float)(c[2 * p + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    float r = __builtin_amdgcn_sqrtf(-2.0f * __logf(u1));      // v_sqrt_f32 (1 ulp) instead of sqrtf's correctly rounded sequence: noise-grade
    // v_sin_f32 / v_cos_f32 take their argument in REVOLUTIONS: sin(2 pi u2) is one instruction each (sincospif: a software range reduction
    // + two polynomials per deviate pair -- a third of this kernel's vector work before round 4); |error| ~1e-6, noise-grade
    out[2 * p] = r * __builtin_amdgcn_cosf(u2);
    out[2 * p + 1] = r * __builtin_amdgcn_sinf(u2);
  }
}
