// Block-scaled fp8 (OCP MX: e4m3 elements, one e8m0 power-of-two scale per 32 elements) operand images of the ViT attention,
// BASELINE config[4] -- written by the attn.qkv GEMM epilogue (vit_ws_gemm.hip) or by the stand-alone quantiser, read by
// vit_attn_mx_kernel.  One caller workspace of maavss_vit_attn_mx_ws_bytes(rows) bytes:
//
//   q8, k8 [rows_alloc][384]        e4m3, row = token (global row index frame * ntok + t), column = head * 64 + d
//   sq, sk [12][rows_alloc]         e8m0, plane = head * 2 + (d >> 5): the scale of the 32 d of that (token, head, half)
//   v8t    [384][rows_alloc]        e4m3, V TRANSPOSED: row = head * 64 + d, column = token -- natural token order
//   sv     [rows_alloc / 32][384]   e8m0, [b][row of v8t]: the scale of the 32 tokens [32 b, 32 b + 32) of that d row: blocks are aligned on the
//                                   GLOBAL row index (frames are ntok = 785 rows apart, not a multiple of 32), which is why the
//                                   attention kernel walks a frame's keys in 32-aligned tiles and masks the rows of its neighbours.
//                                   CONSEQUENCE (documented, tested in tests/test_parity_r4_gpu.py): a block that straddles two frames takes
//                                   ONE scale from both, so a frame's output depends slightly on its neighbours in the launch group (and
//                                   on the zero rows after the last frame) -- unlike the 16-bit modes, which are bit-identical per frame.
//                                   Measured (one frame alone vs inside a group of 8, 224^2): mean map difference 3.3e-3 = 0.44 of the
//                                   mode's own mean distance to fp32; most of it is the tile ALIGNMENT (a frame's 64-key tiles start at
//                                   another key, P' is rounded to e4m3 against another running maximum), not the shared scales.
// rows_alloc = ceil(rows / 128) * 128 + 128: the GEMM stores whole 64-row panels, the attention kernel reads up to 63 rows past
// a frame's last token (masked; the bytes must only be finite e4m3, the stand-alone quantiser / the caller zero the tail).
//
// Why these block shapes: v_mfma_scale_f32_32x32x64_f8f6f4 applies one scale per operand row and 32-element K block.  Q K^T sums
// over d (64 = two blocks per token and head: scales per token), P V sums over keys (blocks of 32 keys: scales per d row and
// 32-token block, constant along the keys inside a block as the instruction requires).
#pragma once
#include "common.h"

#define MX_DIM 384
#define MX_BLOCK 32
#define MX_E4M3_MAX 448.0f

struct MxImages {
  unsigned char *q8, *k8, *sq, *sk, *v8t, *sv;
  int64_t rows_alloc;   // multiple of 128
  int64_t nblk;         // rows_alloc / 32
};

static inline int64_t mx_rows_alloc(int64_t rows) { return (rows + 127) / 128 * 128 + 128; }
static inline int64_t mx_ws_bytes(int64_t rows) {
  const int64_t ra = mx_rows_alloc(rows);
  return 2 * ra * MX_DIM + 2 * 12 * ra + MX_DIM * ra + MX_DIM * (ra / MX_BLOCK) + 256;
}
static inline MxImages mx_images(void* ws, int64_t rows) {
  MxImages m;
  m.rows_alloc = mx_rows_alloc(rows);
  m.nblk = m.rows_alloc / MX_BLOCK;
  unsigned char* p = (unsigned char*)ws;
  m.q8 = p;  p += m.rows_alloc * MX_DIM;
  m.k8 = p;  p += m.rows_alloc * MX_DIM;
  m.v8t = p; p += m.rows_alloc * MX_DIM;
  m.sq = p;  p += 12 * m.rows_alloc;
  m.sk = p;  p += 12 * m.rows_alloc;
  m.sv = p;
  return m;
}

// Scale choice: the OCP MX recipe takes 2^(floor(log2 amax) - 8) and SATURATES block elements between 448 and 512 times the scale;
// this one never saturates -- where the recipe would clip (amax / scale in (448, 512)) it takes the next power of two instead, at the
// price of one bit of resolution for the rest of that block.  Any e8m0 scale is a valid MX encoding; the oracle emulates this one.
// e8m0 scale (biased exponent byte) of a block with absolute maximum `amax`: the smallest power of two s with amax / s <= 448
// (up to the rounding of amax * (1/448): a quotient of 448 (1 + 1e-7) still rounds to 448, e4m3 only overflows above 464).
// amax = 0 -> byte 0 (2^-127): the block's values are exact zeros whatever the scale.  inv = 1 / s as a float (exact).
__device__ __forceinline__ unsigned mx_scale_byte(float amax, float& inv) {
  const unsigned bits = __float_as_uint(amax * (1.0f / MX_E4M3_MAX));
  unsigned e = (bits + 0x7FFFFFu) >> 23;          // ceil to the next power of two (exact powers stay)
  e = e > 253u ? 253u : e;                        // inf / NaN inputs: keep the reciprocal representable (the range guard catches them)
  inv = __uint_as_float((254u - e) << 23);
  return e;
}
__device__ __forceinline__ unsigned mx_cvt4(float a, float b, float c, float d) {
  int w;
  asm volatile("" : "=v"(w));                                // both halves are written below: no zero-initialisation needed
  w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);       // bytes 0, 1
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);        // bytes 2, 3
  return (unsigned)w;
}
