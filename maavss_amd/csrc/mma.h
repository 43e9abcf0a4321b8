// MFMA fragment helpers shared by the GEMM-shaped kernels.
//
// One inner-loop shape serves the three arithmetic modes (include/maavss.h `mode`):
//   MODE 0 (bf16) : LDS holds bf16, one v_mfma_f32_16x16x32_bf16 per 32-deep K step;
//   MODE 2 (f16)  : LDS holds IEEE half, one v_mfma_f32_16x16x32_f16 per step (same rate as bf16, 3 more
//                   mantissa bits: used for forward operands, whose range is bounded by BatchNorm);
//   MODE 1 (f32)  : LDS holds f32, eight v_mfma_f32_16x16x4_f32 per 32-deep K step.  Lane l keeps the
//                   same 8 consecutive k (k = 8*(l>>4)+j) as in the bf16 form; MFMA j consumes element j
//                   of every lane, i.e. the k set {j, 8+j, 16+j, 24+j}; the 8 MFMAs together cover the
//                   32 k exactly once.  Result is an exact-f32 fma chain (guide: FP32-input MFMA).
// Operand tiles in LDS are [row][32 k] (k contiguous) for both the M-side and the N-side operand.
// C/D layout of a 16x16 tile: col = lane & 15, row = (lane >> 4) * 4 + reg.
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;

#define MODE_BF16 0
#define MODE_F32 1
#define MODE_F16 2

typedef __attribute__((ext_vector_type(8))) _Float16 mfma_f16x8;

template <int MODE>
struct Mma;

template <>
struct Mma<MODE_F16> {
  using elem = unsigned short;  // raw IEEE half bits
  using frag = bf16x8;
  static __device__ __forceinline__ elem cvt(float f) {
    _Float16 h = (_Float16)f;
    return __builtin_bit_cast(unsigned short, h);
  }
  static __device__ __forceinline__ frag load(const elem* p) { return *reinterpret_cast<const bf16x8*>(p); }
  static __device__ __forceinline__ void mma(f32x4& acc, const frag& a, const frag& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mfma_f16x8, a), __builtin_bit_cast(mfma_f16x8, b),
                                                 acc, 0, 0, 0);
  }
};

template <>
struct Mma<MODE_BF16> {
  using elem = bf16_t;
  using frag = bf16x8;
  static __device__ __forceinline__ elem cvt(float f) { return f2bf(f); }
  static __device__ __forceinline__ frag load(const elem* p) { return *reinterpret_cast<const bf16x8*>(p); }
  static __device__ __forceinline__ void mma(f32x4& acc, const frag& a, const frag& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                  __builtin_bit_cast(mfma_bf16x8, b), acc, 0, 0, 0);
  }
};

template <>
struct Mma<MODE_F32> {
  using elem = float;
  struct frag {
    f32x4 lo, hi;
  };
  static __device__ __forceinline__ elem cvt(float f) { return f; }
  static __device__ __forceinline__ frag load(const elem* p) {
    frag f;
    f.lo = *reinterpret_cast<const f32x4*>(p);
    f.hi = *reinterpret_cast<const f32x4*>(p + 4);
    return f;
  }
  static __device__ __forceinline__ void mma(f32x4& acc, const frag& a, const frag& b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.lo[j], b.lo[j], acc, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.hi[j], b.hi[j], acc, 0, 0, 0);
  }
};

// 32x32x16 forms (f32x16 accumulator: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5); A/B: lane (r = lane & 31,
// h = lane >> 5) holds A[row r][k = 8h + j] / B[k = 8h + j][col r], j = 0..7).  Half the vector-issue hold per FLOP of the
// 16x16x32 forms (8 of 32 cycles instead of 8 of 16).
template <int MODE>
struct Mma32;
template <>
struct Mma32<MODE_BF16> {
  static __device__ __forceinline__ void mma(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a), __builtin_bit_cast(mfma_bf16x8, b), acc, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mma3(const bf16x8& a, const bf16x8& b, const f32x16& c) {   // D = A B + C, C kept
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a), __builtin_bit_cast(mfma_bf16x8, b), c, 0, 0, 0);
  }
};
template <>
struct Mma32<MODE_F16> {
  static __device__ __forceinline__ void mma(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(mfma_f16x8, a), __builtin_bit_cast(mfma_f16x8, b), acc, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mma3(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(mfma_f16x8, a), __builtin_bit_cast(mfma_f16x8, b), c, 0, 0, 0);
  }
};
