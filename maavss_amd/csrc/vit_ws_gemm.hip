// K2/K3, weight-stationary form: the K = 384 dense layers of the ViT blocks (attn.qkv, attn.proj, mlp.fc1 of dino's Block,
// reached from the reference through video_attention.py:52) with the WEIGHTS held in registers and the activations streamed.
//
// Why: the panel-stationary kernel (vit_panel_gemm.hip) keeps a [128 x 384] activation panel in LDS and streams weight tiles
// through an LDS ring; per 32-deep step its eight MFMA waves read 48 KiB of fragments for 258 cycles of MFMA (75 % of the
// LDS bandwidth), the panel build and the output stores run with the matrix pipe idle, and the ring takes the other 64 KiB.
// Here
//   * a workgroup (12 waves, 3 per SIMD) owns 384 output columns: wave w keeps W[n0 + 32 w .. +32][0 .. 384) as 24
//     A-operand fragments of v_mfma_f32_32x32x16 (96 registers) for its whole life -- no weight traffic after the first
//     25 KiB per wave;
//   * the 16-bit activation rows stream through LDS in [64 x 384] panels (48 KiB, two buffers): each wave fetches 4 KiB of
//     the panel after next into registers (four 16-byte loads, in flight during a whole panel of MFMAs) and writes them to
//     the free buffer after the next barrier.  (LDS-DMA would save the 16 registers, but hipcc waits vmcnt(0) before every
//     LDS read of a wave that has a global_load_lds in flight -- it cannot tell the destinations apart.)  Every wave reads
//     the whole panel: one 16-byte fragment read per MFMA = 50 % of the LDS read bandwidth at full matrix rate;
//   * N > 384 (qkv 1152, fc1 1536) is covered by ns = N / 384 workgroups that walk the SAME rows on CUs of one XCD (the
//     panel is fetched from HBM once and served to the others by that XCD's L2);
//   * each wave stores its own [32 x 32] outputs straight from the accumulators (the MFMA row permutation below gives a lane
//     8 consecutive columns): with three waves per SIMD one wave's epilogue runs under the other two's MFMAs;
//   * one workgroup barrier per panel.
// Epilogues: 0 +bias, q-scale -> 16-bit | 1 +bias, GELU -> 16-bit (4: the same with the polynomial evaluated in packed IEEE half -- faster, twice
// the rounding error of the stored value: a selectable mode, vit_epilogue.h) | 2 +bias +residual -> f32 in place, and optionally the
// NEXT LayerNorm of the updated rows -> 16-bit (ns = 1: the workgroup holds whole rows), so that the consumer GEMM needs no
// LayerNorm pass | 3 (round 3, attn.qkv only, N = 1152): +bias, q-scale -> the BLOCK-SCALED fp8 operand images of the attention
// kernel (vit_mx.h) instead of a 16-bit qkv tensor: the q and k slices per (token, 32 columns) -- a wave's 32 output columns are
// one scale block, absmax over a lane's 16 values + one lane swap --, the v slice TRANSPOSED: its workgroups issue the MFMA with
// the operands exchanged (weights as B), so a lane holds one d column and 16 tokens of a 32-token block, quantises per (d, block)
// and stores 16 consecutive token bytes of a V^T row.  No quantisation pass, 1 byte per element written instead of 2.
#include <type_traits>
#include <stdlib.h>
#include "mma.h"
#include "vit_epilogue.h"
#include "vit_mx.h"

#define WS_K 384
#define WS_BM 64
#define WS_WAVES 12
#define WS_THREADS (WS_WAVES * 64)
#define WS_SLICE (WS_WAVES * 32)
#define WS_BUFS 2
#define WS_PANEL_BYTES (WS_BM * WS_K * 2)

struct WsArgs {
  const bf16_t* A;       // 16-bit [ceil(M/64)*64][384]
  const float* X;        // LN = 2 instead of A: f32 [ceil(M/64)*64][384], normalised on the way into LDS
  const float* row_stats;  // LN = 2: [M][3][2] = (mean, sum of squared deviations) of the three 128-column thirds of each row of X
  const bf16_t* W;       // 16-bit [N][384]
  const float* bias;     // [N]
  void* C;               // 16-bit [c_rows][ldc] (epi 0, 1) or f32 [c_rows][ldc] (epi 2, read-modify-write)
  bf16_t* XN;            // epi 2, optional: LayerNorm(C row) -> 16-bit [c_rows][384]
  const float* ln_g;
  const float* ln_b;
  float ln_eps;
  int M, N, ldc;
  int qscale_cols;
  float qscale;
  int ns;                // column slices of 384
  int groups_per_xcd;    // (CUs per XCD) / ns
  int panels;            // ceil(M / 64)
  MxImages mx;           // epi 3
};

// SH = MFMA shape: 32 = v_mfma_f32_32x32x16 (rounds 2-3; kept for the MX epilogue and as the measured baseline), 16 = v_mfma_f32_16x16x32
// (round 4, default).  Same FLOPs per cycle on paper; on RANDOM operands the chip is power-limited and holds a higher clock on the
// 16x16x32 form (scripts/valu_probe part 3: 1612 -> 1868 TFLOP/s, +16 %, register-resident f16 loops; both 2470 on zeros), and a
// 16x16 accumulator gives a lane 8 consecutive output columns of ONE row with the four lanes of a row adjacent: 64 contiguous
// bytes per row and store instruction instead of 32.
#ifdef WS_STAMP
// measurement build (scripts/ws_stamp.py; never loaded by the package): where a wave's cycles go, per phase, summed over its panels
__device__ unsigned long long ws_stamps[256 * WS_WAVES * 8];
extern "C" int maavss_ws_stamps_read(unsigned long long* host, int clear) {
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(ws_stamps), sizeof(ws_stamps)) != hipSuccess) return 1;
  if (clear) { static unsigned long long z[256 * WS_WAVES * 8]; if (hipMemcpyToSymbol(HIP_SYMBOL(ws_stamps), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#define WS_T(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_last; st_last = t_; }
#else
#define WS_T(i)
#endif

template <int EPI, int LN, int MODE, int SH>
__global__ __launch_bounds__(WS_THREADS) void vit_ws_gemm_kernel(WsArgs g) {
  static_assert(SH == 32 || SH == 16, "MFMA shape");
  static_assert(SH == 32 || EPI != 3, "the MX epilogue is built on the 32x32 accumulator layout");
  // LN: 0 none | 1 LayerNorm of the output rows (proj -> norm2) | 2 LayerNorm of the input rows on the way in (norm1 -> qkv) | 3 / 4 = 1 / 2 WITHOUT
  // the affine part (ln_gamma = ln_beta = null: plain (x - mean) * rstd), for callers that fold gamma into the consumer GEMM's frozen weight columns
  // and beta into its bias (W' = W diag(gamma), b' = b + W beta): the deposit of norm1-on-the-way-in then costs two vector instructions per element
  // instead of three and no gamma / beta reads from LDS.  Measured in the step: +0.3 % (HISTORY.md H4) -- VideoAttention does not fold by default.
  // LN = 5 (round 4, "LayerNorm after the product"): the rows arrive as f32 like LN = 2 / 4 but are only ROUNDED on the way in; the weights are the
  // gamma-folded W' = W diag(gamma), and the epilogue applies the row statistics:  y = rstd_r (x W'^T - mean_r s_n) + b'_n  with s_n = sum_k W'[n][k]
  // (g.ln_g, one per output column) and b' = b + W beta (g.bias) -- identical to LayerNorm(x) W^T + b in exact arithmetic.  The deposit is two
  // conversions per four elements instead of the normalisation (which costs the step's attn.qkv launches a third of their time).
  constexpr bool LN_POST = LN == 5;
  constexpr bool LN_OUT = LN == 1 || LN == 3, LN_IN = LN == 2 || LN == 4 || LN_POST, AFF = LN == 1 || LN == 2;
  static_assert(LN >= 0 && LN <= 5, "LN variant");
  static_assert(!LN_POST || (SH == 16 && EPI == 0), "LayerNorm after the product: 16x16x32 form, 16-bit epilogue");
  static_assert(!LN_OUT || EPI == 2, "LayerNorm of the output rows comes with the residual epilogue");
  static_assert(!LN_IN || EPI == 0 || EPI == 3, "LayerNorm on the way in is wired for the attn.qkv epilogues");
  static_assert(EPI != 3 || LN_IN, "the MX epilogue is wired for attn.qkv (LayerNorm on the way in)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* panels_lds = smem;                                                       // 2 x 48 KiB
  float* bias_lds = reinterpret_cast<float*>(smem + WS_BUFS * WS_PANEL_BYTES);   // [384] (+ [384] gamma, [384] beta, stats)
  float* gam_lds = bias_lds + WS_SLICE;
  float* bet_lds = gam_lds + WS_SLICE;
  float2* stat_lds = reinterpret_cast<float2*>(bet_lds + WS_SLICE);              // [12 waves][64 rows] (mean, M2) of 32 columns
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r32 = lane & 31, half = lane >> 5;
  // workgroup -> (XCD, slot): dispatch is round-robin over the 8 XCDs, so blockIdx & 7 is the XCD
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  if (slot >= g.groups_per_xcd * g.ns) return;
  const int slice = slot % g.ns, group = xcd * g.groups_per_xcd + slot / g.ns, ngroups = 8 * g.groups_per_xcd;
  const int p0 = (int)((int64_t)g.panels * group / ngroups), p1 = (int)((int64_t)g.panels * (group + 1) / ngroups);
  if (p0 >= p1) return;
  const int n0 = slice * WS_SLICE + wv * 32;
#ifdef MAAVSS_WS_ABL
  constexpr int abl = MAAVSS_WS_ABL;   // measurement builds only (scripts/gemm_ablate.sh): bit mask of loop parts to skip
#else
  constexpr int abl = 0;
#endif

  // ---- panel stream.  A panel is 48 KiB of contiguous memory = 3072 chunks of 16 B; chunk q = 64 (4 wv + i) + lane is row
  // q / 48, chunk c = q % 48 of it, and goes to LDS byte r * 768 + ((c ^ (r & 15)) << 4): a fragment read (32 rows x 2
  // chunks per wave-instruction) then touches every bank group once per 16-lane service group.
  // The four loads are inline asm and their completion is waited for by hand (WS_WAIT_FETCH): vmcnt retires in issue order
  // over loads AND stores, and hipcc, merging the loop-carried state, waits vmcnt(0..3) before the LDS writes -- i.e. for the
  // acknowledgement of the output stores issued just before -- and flushes vmcnt(0) in front of the inner loop.  Counted by
  // hand, the wait lets exactly the `S` vector-memory operations of one panel epilogue (all younger than the fetch) fly.
  // The compiler does not know that the four registers are pending between fetch and wait: both sit in the SAME loop
  // iteration (no loop-carried copy), the wait has no register operands (no tied-operand copy), and the build runs
  // check_ws_gemm_isa.py over the emitted ISA, which fails if any instruction touches them in between or a kernel spills.
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 stg0, stg1, stg2, stg3;
#define WS_FETCH(p)                                                                                               \
  {                                                                                                               \
    const char* src_ = reinterpret_cast<const char*>(g.A + (int64_t)(p) * (WS_BM * WS_K)) + (wv * 256 + lane) * 16; \
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:1024\n\t"            \
                 "global_load_dwordx4 %2, %4, off offset:2048\n\tglobal_load_dwordx4 %3, %4, off offset:3072"      \
                 : "=&v"(stg0), "=&v"(stg1), "=&v"(stg2), "=&v"(stg3) : "v"(src_) : "memory");                     \
  }
#define WS_WAIT_FETCH(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory");
#define WS_DEPOSIT_ONE(i, v)                                                                                      \
  {                                                                                                               \
    const int q_ = (wv * 4 + i) * 64 + ln_, r_ = q_ / 48, c_ = q_ - r_ * 48;                                      \
    *reinterpret_cast<u32x4*>(dst_ + r_ * (WS_K * 2) + ((c_ ^ (r_ & 15)) << 4)) = v;                             \
  }
#define WS_DEPOSIT(buf)                                                                                           \
  {                                                                                                               \
    int ln_ = lane;                                                                                               \
    asm volatile("" : "+v"(ln_)); /* keeps the four offsets out of the loop-invariant (= permanently live) registers */ \
    char* dst_ = panels_lds + (buf) * WS_PANEL_BYTES;                                                             \
    WS_DEPOSIT_ONE(0, stg0) WS_DEPOSIT_ONE(1, stg1) WS_DEPOSIT_ONE(2, stg2) WS_DEPOSIT_ONE(3, stg3)               \
  }
  // residual epilogues (HBM-bound, and short of registers once the LayerNorm output is on): the panel travels in two halves of
  // 32 rows, each fetched at the start and written at the end of the half-panel step it overlaps -- 8 staging registers
#define WS_FETCH_HALF(p, part)                                                                                    \
  {                                                                                                               \
    const char* src_ = reinterpret_cast<const char*>(g.A + (int64_t)(p) * (WS_BM * WS_K)) + ((part) * 1536 + wv * 128 + lane) * 16; \
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:1024"                 \
                 : "=&v"(stg0), "=&v"(stg1) : "v"(src_) : "memory");                                               \
  }
#define WS_DEPOSIT_HALF_ONE(part, i, v)                                                                           \
  {                                                                                                               \
    const int q_ = (part) * 1536 + (wv * 2 + i) * 64 + ln_, r_ = q_ / 48, c_ = q_ - r_ * 48;                      \
    *reinterpret_cast<u32x4*>(dst_ + r_ * (WS_K * 2) + ((c_ ^ (r_ & 15)) << 4)) = v;                              \
  }
#define WS_DEPOSIT_HALF(buf, part)                                                                                \
  {                                                                                                               \
    int ln_ = lane;                                                                                               \
    asm volatile("" : "+v"(ln_));                                                                                 \
    char* dst_ = panels_lds + (buf) * WS_PANEL_BYTES;                                                             \
    WS_DEPOSIT_HALF_ONE(part, 0, stg0) WS_DEPOSIT_HALF_ONE(part, 1, stg1)                                         \
  }
  // LN = 2 (norm1 -> attn.qkv): the rows arrive as f32 and are LayerNorm-ed on the way into LDS.  A half panel (32 rows) of f32
  // is as many bytes as a whole 16-bit panel: the same four loads per lane, one half per half-panel step.  Row statistics:
  // the producer of X (vit_gemm's f32 epilogues) left (mean, M2) of each 128-column third; lane l loads the three pairs of
  // row l of the panel AFTER next together with the first half's fetch (same hand-counted wait), merges them (parallel-variance
  // formula) and wave 0 puts (mean, rstd) into a three-slot LDS table, which the next panel barrier publishes.
  float4 st01;
  float2 st2;
#define WS_FETCH_X(hp, with_stats, srow)                                                                           \
  {                                                                                                               \
    const char* src_ = reinterpret_cast<const char*>(g.X) + (int64_t)(hp) * WS_PANEL_BYTES + (wv * 256 + lane) * 16; \
    if (with_stats) {                                                                                             \
      const char* sp_ = reinterpret_cast<const char*>(g.row_stats) + (int64_t)(srow) * 24;                         \
      asm volatile("global_load_dwordx4 %0, %6, off\n\tglobal_load_dwordx4 %1, %6, off offset:1024\n\t"            \
                   "global_load_dwordx4 %2, %6, off offset:2048\n\tglobal_load_dwordx4 %3, %6, off offset:3072\n\t" \
                   "global_load_dwordx4 %4, %7, off\n\tglobal_load_dwordx2 %5, %7, off offset:16"                  \
                   : "=&v"(stg0), "=&v"(stg1), "=&v"(stg2), "=&v"(stg3), "=&v"(st01), "=&v"(st2) : "v"(src_), "v"(sp_) : "memory"); \
    } else {                                                                                                      \
      asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:1024\n\t"            \
                   "global_load_dwordx4 %2, %4, off offset:2048\n\tglobal_load_dwordx4 %3, %4, off offset:3072"    \
                   : "=&v"(stg0), "=&v"(stg1), "=&v"(stg2), "=&v"(stg3) : "v"(src_) : "memory");                   \
    }                                                                                                             \
  }
  float2* table_lds = stat_lds;      // [3][64] (mean, rstd); LN = 1 uses the same bytes for its partials
  auto merge_stats = [&](const float4& a, const float2& b) __attribute__((always_inline)) {
    const float mean = (a.x + a.z + b.x) * (1.f / 3.f);
    const float d0 = a.x - mean, d1 = a.z - mean, d2 = b.x - mean;
    const float m2 = (a.y + a.w + b.y) + 128.f * (d0 * d0 + d1 * d1 + d2 * d2);
    return make_float2(mean, rsqrtf(m2 * (1.f / WS_K) + g.ln_eps));
  };
  // chunk q = 64 (4 wv + i) + lane of a half = 4 floats of row q / 96 (+ 32 part), columns 4 (q % 96)...: normalise, round,
  // 8 bytes to LDS (same swizzled image as the 16-bit path)
  auto deposit_ln_one = [&](char* dst, const float2* tab, int part, int i, const u32x4& raw, int ln) __attribute__((always_inline)) {
    const int q = (wv * 4 + i) * 64 + ln, r = q / 96, c4 = q - r * 96, rr = 32 * part + r;
    float2 ms = make_float2(0.f, 0.f);
    if constexpr (!LN_POST) ms = tab[rr];
    const float4 v = __builtin_bit_cast(float4, raw);
    uint2 o;
    if constexpr (LN_POST) {
      o = make_uint2(pack2<MODE>(v.x, v.y), pack2<MODE>(v.z, v.w));
    } else if constexpr (AFF) {
      const float4 gg = *reinterpret_cast<const float4*>(gam_lds + 4 * c4), bb = *reinterpret_cast<const float4*>(bet_lds + 4 * c4);
      o = make_uint2(pack2<MODE>((v.x - ms.x) * ms.y * gg.x + bb.x, (v.y - ms.x) * ms.y * gg.y + bb.y),
                     pack2<MODE>((v.z - ms.x) * ms.y * gg.z + bb.z, (v.w - ms.x) * ms.y * gg.w + bb.w));
    } else {
      o = make_uint2(pack2<MODE>((v.x - ms.x) * ms.y, (v.y - ms.x) * ms.y), pack2<MODE>((v.z - ms.x) * ms.y, (v.w - ms.x) * ms.y));
    }
    *reinterpret_cast<uint2*>(dst + rr * (WS_K * 2) + (((c4 >> 1) ^ (rr & 15)) << 4) + (c4 & 1) * 8) = o;
  };
#define WS_DEPOSIT_LN(buf, tab, part)                                                                             \
  {                                                                                                               \
    int ln_ = lane;                                                                                               \
    asm volatile("" : "+v"(ln_));                                                                                 \
    char* dst_ = panels_lds + (buf) * WS_PANEL_BYTES;                                                             \
    deposit_ln_one(dst_, tab, part, 0, stg0, ln_); deposit_ln_one(dst_, tab, part, 1, stg1, ln_);                 \
    deposit_ln_one(dst_, tab, part, 2, stg2, ln_); deposit_ln_one(dst_, tab, part, 3, stg3, ln_);                 \
  }
  constexpr bool SPLIT = EPI == 2;
  if constexpr (!LN_IN) WS_FETCH(p0)

  // ---- stationary weights.  MFMA row i = 8 q + 4 h + r of the A operand is fed with weight row n0 + pi(i),
  // pi(i) = 16 (q >> 1) + 8 h + 4 (q & 1) + r: the accumulator registers (q, r) of a lane (column = activation row, h = lane
  // half) are then the 8 consecutive output columns 16 (q >> 1) + 8 h + (0..7) for q = (0,1) and (2,3).
  // SH = 16: A-operand block b (16 weight rows) x k-step s (32 deep) = w[12 b + s]; lane (i = lane & 15, G = lane >> 4) holds k = 32 s + 8 G ..
  // of weight row n0 + col(b, i).  The accumulator of block b gives lane (j, G) the rows 4 G + r, so
  //   16-bit outputs: col(b, i) = 8 (i >> 2) + 4 b + (i & 3)  ->  the lane's 8 values are the columns 8 G + (0..7): one 16-byte store, and
  //                   the four lanes G = 0..3 of activation row j cover 64 contiguous bytes;
  //   f32 outputs:    col(b, i) = 16 b + i                    ->  block b gives the lane columns 16 b + 4 G + (0..3): one float4, the four
  //                   lanes of a row again 64 contiguous bytes per instruction.
  // (with the LayerNorm output the 16-bit row wants the 8-consecutive layout: one 16-byte store and 20 instead of 24 vector-memory
  // operations per panel; measured 429 -> see DESIGN.md; WS16_LN_SPLITCOLS=1 restores the split layout for A/B)
#ifndef WS16_LN_SPLITCOLS
#define WS16_LN_SPLITCOLS 0
#endif
  constexpr bool F32OUT = EPI == 2 && (!LN_OUT || WS16_LN_SPLITCOLS);
  constexpr int CSTEP = F32OUT ? 16 : 4;              // column distance between the lane's block-0 and block-1 values
  const int j16 = lane & 15, g16 = lane >> 4;
  const int cbase16 = F32OUT ? 4 * g16 : 8 * g16;     // first column (within the wave's 32) of the lane's block-0 values
  bf16x8 w[24];
  if constexpr (SH == 32) {
    const int q = r32 >> 3, h = (r32 >> 2) & 1, r = r32 & 3;
    const bf16_t* wp = g.W + (int64_t)(n0 + 16 * (q >> 1) + 8 * h + 4 * (q & 1) + r) * WS_K + 8 * half;
#pragma unroll
    for (int kb = 0; kb < 24; ++kb) w[kb] = *reinterpret_cast<const bf16x8*>(wp + kb * 16);
  } else {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = F32OUT ? 16 * b + j16 : 8 * (j16 >> 2) + 4 * b + (j16 & 3);
      const bf16_t* wp = g.W + (int64_t)(n0 + col) * WS_K + 8 * g16;
#pragma unroll
      for (int ks = 0; ks < 12; ++ks) w[12 * b + ks] = *reinterpret_cast<const bf16x8*>(wp + ks * 32);
    }
  }
  if (tid < WS_SLICE) {
    bias_lds[tid] = g.bias[slice * WS_SLICE + tid];
    if constexpr (AFF) { gam_lds[tid] = g.ln_g[tid]; bet_lds[tid] = g.ln_b[tid]; }
    if constexpr (LN_POST) gam_lds[tid] = g.ln_g[slice * WS_SLICE + tid];      // s_n of this workgroup's 384 output columns
  }
  if constexpr (!LN_IN) {
    WS_WAIT_FETCH(0)
    WS_DEPOSIT(0)
  } else {
    // statistics of the first two panels (plain loads), then the first panel half by half
    if (wv < 2 && p0 + wv < p1) {
      int row = (p0 + wv) * WS_BM + lane;
      row = row < g.M ? row : g.M - 1;
      const float* sp = g.row_stats + (int64_t)row * 6;
      table_lds[wv * WS_BM + lane] = merge_stats(*reinterpret_cast<const float4*>(sp), *reinterpret_cast<const float2*>(sp + 4));
    }
    __syncthreads();
#pragma unroll 1
    for (int part = 0; part < 2; ++part) {
      WS_FETCH_X(2 * p0 + part, false, 0)
      WS_WAIT_FETCH(0)
      WS_DEPOSIT_LN(0, table_lds, part)
    }
  }
  // fragment read address: activation row r32 (+ 32 h2), chunk 2 kb + half, kb = 8 a + b:
  //   ((2 b + half) ^ (r32 & 15)) << 4  =  (((b << 5) ^ ((r32 & 14) << 4))) + ((half ^ (r32 & 1)) << 4)
  const int frag_r = r32 * (WS_K * 2) + ((half ^ (r32 & 1)) << 4), frag_x = (r32 & 14) << 4;
  // SH = 16: activation row 16 c + j16, chunk 4 s + g16:  ((4 s + g16) ^ j16) << 4  =  ((g16 ^ (j16 & 3)) << 4) + (((s & 3) << 6) ^ ((j16 & 12) << 4)) + (s >> 2) * 256
  const int frag16_r = j16 * (WS_K * 2) + ((g16 ^ (j16 & 3)) << 4), frag16_x = (j16 & 12) << 4;
  const float scale = ((EPI == 0 || EPI == 3) && slice * WS_SLICE < g.qscale_cols) ? g.qscale : 1.f;   // qscale_cols is a multiple of 384
  const bool vslice = EPI == 3 && slice == 2;          // attn.qkv -> MX images: the v slice runs with the MFMA operands exchanged

  // vector-memory operations of one panel epilogue (loads + stores) that the hand-counted wait may leave in flight.  EPI 3: a
  // half-panel issues 3 (q, k slices: two 8-byte stores + the scale byte) or 2 (v slice: one 16-byte store + the scale byte)
  // (SH = 16 with the LayerNorm output: the 16-bit row goes out as two 8-byte pieces per 16-row block instead of one 16-byte piece)
  constexpr int S = EPI == 2 ? (LN_OUT ? (SH == 16 && WS16_LN_SPLITCOLS ? 24 : 20) : 16) : 4;
  int it = 0;
#ifdef WS_STAMP
  unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#endif
  for (int p = p0; p < p1; ++p, ++it) {
    const int buf = it & 1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(abl & 16)) __builtin_amdgcn_s_barrier();      // panel p is in LDS for everybody; everybody is done reading panel p-1
    WS_T(0)                                             // [0] waiting at the panel barrier
    const bool more = p + 1 < p1 && !((abl & 2) && it > 0);
    if constexpr (!SPLIT && !LN_IN) { if (more) WS_FETCH(p + 1) }   // in flight during this panel's MFMAs, written to the other buffer at the end
    const int m0 = p * WS_BM;
#pragma unroll 1
    for (int h2 = 0; h2 < 2; ++h2) {
      if constexpr (SPLIT) { if (more) WS_FETCH_HALF(p + 1, h2) }
      bool stats_next = false;
      if constexpr (LN_IN) {
        if (more) {
          stats_next = h2 == 0 && p + 2 < p1;
          int srow = (p + 2) * WS_BM + lane;
          srow = srow < g.M ? srow : g.M - 1;
          WS_FETCH_X(2 * (p + 1) + h2, stats_next, srow)
        }
      }
      if constexpr (SH == 16) {
        // ================= 16x16x32 form: two 16-row blocks (c) x two 16-column blocks (b) of accumulators =================
        const char* pb16 = panels_lds + buf * WS_PANEL_BYTES + h2 * (32 * WS_K * 2) + frag16_r;
        f32x4 acc[2][2];
        if constexpr (LN_POST) {
          const float2* tab16 = table_lds + (it % 3) * WS_BM + 32 * h2 + j16;      // (mean, rstd) of this lane's rows (c = 0: [0], c = 1: [16])
          const float nm0 = -tab16[0].x, nm1 = -tab16[16].x;
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const float4 sq = *reinterpret_cast<const float4*>(gam_lds + wv * 32 + cbase16 + CSTEP * b);
            acc[0][b] = f32x4{nm0 * sq.x, nm0 * sq.y, nm0 * sq.z, nm0 * sq.w};      // - mean_r s_n: the MFMAs add x W'^T
            acc[1][b] = f32x4{nm1 * sq.x, nm1 * sq.y, nm1 * sq.z, nm1 * sq.w};
          }
        } else {
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const float4 bq = *reinterpret_cast<const float4*>(bias_lds + wv * 32 + cbase16 + CSTEP * b);
            acc[0][b] = f32x4{bq.x, bq.y, bq.z, bq.w};
            acc[1][b] = acc[0][b];
          }
        }
        // fragment t = 2 ks + c: rows 16 c .., k-step ks; it feeds the two MFMAs of column blocks b = 0, 1
        auto rd16 = [&](int t) __attribute__((always_inline)) {
          int fx = frag16_x;
          if constexpr (EPI != 1) asm volatile("" : "+v"(fx));
          const int c = t & 1, ks = t >> 1;
          return *reinterpret_cast<const bf16x8*>(pb16 + c * (16 * WS_K * 2) + (((ks & 3) << 6) ^ fx) + (ks >> 2) * 256);
        };
        constexpr int D16 = LN == 4 ? 2 : LN_IN ? 1 : (EPI == 2) ? 2 : 3;       // fragments in flight ahead of their MFMAs (the variants short of registers: 2)
        {
          bf16x8 f[D16 + 1];
#pragma unroll
          for (int t = 0; t < D16; ++t) f[t] = rd16(t);
#pragma unroll
          for (int t = 0; t < 24; ++t) {
            if (t + D16 < 24 && !(abl & 8)) f[(t + D16) % (D16 + 1)] = rd16(t + D16);
            __builtin_amdgcn_sched_barrier(0);
            if (!(abl & 4)) {
              Mma<MODE>::mma(acc[t & 1][0], w[t >> 1], f[t % (D16 + 1)]);
              Mma<MODE>::mma(acc[t & 1][1], w[12 + (t >> 1)], f[t % (D16 + 1)]);
            } else if (t == 23) {
#pragma unroll
              for (int i = 0; i <= D16; ++i) acc[0][0][0] += (float)f[i][i];
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        WS_T(1)                                         // [1] bias init + fragment reads + 48 MFMAs
        // ---- epilogue: lane (j16, g16) holds, of rows m0 + 32 h2 + 16 c + j16 (c = 0, 1): 16-bit outputs -- the columns n0 + 8 g16 + (0..7)
        // = (acc[c][0], acc[c][1]); f32 outputs -- the columns n0 + 16 b + 4 g16 + (0..3) = acc[c][b]
        const int64_t row0 = m0 + 32 * h2 + j16;
        bool skip_epi16 = false;
        if constexpr ((abl & 1) != 0) skip_epi16 = acc[0][0][3] != 1234.5f;
        if (!skip_epi16) {
        if constexpr (EPI != 2) {
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            bf16_t* cp = reinterpret_cast<bf16_t*>(g.C) + (row0 + 16 * c) * g.ldc + n0 + 8 * g16;
            v2f v[4] = {v2f{acc[c][0][0], acc[c][0][1]}, v2f{acc[c][0][2], acc[c][0][3]}, v2f{acc[c][1][0], acc[c][1][1]}, v2f{acc[c][1][2], acc[c][1][3]}};
            if constexpr (LN_POST) {      // rstd_r (x W'^T - mean_r s_n) + b'_n, scalar fmas (vit_epilogue.h); addresses redone here: nothing live across the MFMAs
              int jj = lane;
              asm volatile("" : "+v"(jj));
              const float rs = table_lds[(it % 3) * WS_BM + 32 * h2 + 16 * c + (jj & 15)].y;
              const float* bp = bias_lds + (tid >> 6) * 32 + 8 * (jj >> 4);
              {
                const float4 b0 = *reinterpret_cast<const float4*>(bp);
                v[0].x = v[0].x * rs + b0.x; v[0].y = v[0].y * rs + b0.y; v[1].x = v[1].x * rs + b0.z; v[1].y = v[1].y * rs + b0.w;
              }
              {
                const float4 b1 = *reinterpret_cast<const float4*>(bp + 4);
                v[2].x = v[2].x * rs + b1.x; v[2].y = v[2].y * rs + b1.y; v[3].x = v[3].x * rs + b1.z; v[3].y = v[3].y * rs + b1.w;
              }
            }
            if constexpr (EPI == 4) {
              *reinterpret_cast<uint4*>(cp) = make_uint4(pg_gelu_h2(v[0]), pg_gelu_h2(v[1]), pg_gelu_h2(v[2]), pg_gelu_h2(v[3]));
              continue;
            } else if constexpr (EPI == 1) {
              pg_gelu4(v[0], v[1]);
              pg_gelu4(v[2], v[3]);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[e].x *= scale; v[e].y *= scale; }     // scalar multiplies on purpose (vit_epilogue.h)
            }
            *reinterpret_cast<uint4*>(cp) = make_uint4(pack2<MODE>(v[0].x, v[0].y), pack2<MODE>(v[1].x, v[1].y),
                                                       pack2<MODE>(v[2].x, v[2].y), pack2<MODE>(v[3].x, v[3].y));
          }
        } else {
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            float* cp = reinterpret_cast<float*>(g.C) + (row0 + 16 * c) * g.ldc + n0 + cbase16;
            const float4 x0 = *reinterpret_cast<const float4*>(cp), x1 = *reinterpret_cast<const float4*>(cp + CSTEP);
            acc[c][0][0] += x0.x; acc[c][0][1] += x0.y; acc[c][0][2] += x0.z; acc[c][0][3] += x0.w;     // scalar adds on purpose (vit_epilogue.h)
            acc[c][1][0] += x1.x; acc[c][1][1] += x1.y; acc[c][1][2] += x1.z; acc[c][1][3] += x1.w;
            *reinterpret_cast<float4*>(cp) = make_float4(acc[c][0][0], acc[c][0][1], acc[c][0][2], acc[c][0][3]);
            *reinterpret_cast<float4*>(cp + CSTEP) = make_float4(acc[c][1][0], acc[c][1][1], acc[c][1][2], acc[c][1][3]);
          }
          if constexpr (LN_OUT) {
            // LayerNorm of the updated rows.  Per wave: (mean, M2) of its 32 columns of each of the 32 rows.  A lane holds 8 of them for
            // row (c = 0, j16) and 8 for row (c = 1, j16); the other 24 are in the lanes j16 + 16 G'.  ONE pair of lane swaps reduces both
            // rows at once: permlane16_swap on (s0, s1) leaves a + b = (s0_0 + s0_1, s1_0 + s1_1, s0_2 + s0_3, s1_2 + s1_3) by 16-lane row,
            // permlane32_swap on two copies of that adds rows 0 + 2 and 1 + 3 -> lanes with even G hold row c = 0's total, odd G row c = 1's.
            auto reduce2 = [&](float s0, float s1) __attribute__((always_inline)) {
              asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(s0), "+v"(s1));
              float ua, ub;
              lane_swap32(s0 + s1, ua, ub);
              return ua + ub;
            };
            const int cme = g16 & 1;                              // the row block this lane merges: rows 16 cme + j16
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
              for (int r = 0; r < 4; ++r) { s0 += acc[0][b][r]; s1 += acc[1][b][r]; }
            const float mu = reduce2(s0, s1) * (1.f / 32.f);       // wave-level mean of row (cme, j16)
            float mu0, mu1;
            lane_swap16(mu, mu0, mu1);                             // rows (v0, v0, v2, v2) / (v1, v1, v3, v3): both rows' means in every lane
            float q0 = 0.f, q1 = 0.f;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float d0 = acc[0][b][r] - mu0, d1 = acc[1][b][r] - mu1;
                q0 += d0 * d0;
                q1 += d1 * d1;
              }
            const float m2 = reduce2(q0, q1);
            // [wave][row] partials (conflict-free: consecutive 8-byte slots per 16-lane group)
            float2* sp = stat_lds + (32 * h2 + 16 * cme + j16);
            if (g16 < 2) sp[wv * WS_BM] = make_float2(mu, m2);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            float mean = 0.f;
#pragma unroll 4
            for (int k = 0; k < WS_WAVES; ++k) mean += sp[k * WS_BM].x;
            mean *= 1.f / WS_WAVES;
            float tot = 0.f;
#pragma unroll 4
            for (int k = 0; k < WS_WAVES; ++k) { const float2 st = sp[k * WS_BM]; const float d = st.x - mean; tot += st.y + 32.f * d * d; }
            const float rstd = rsqrtf(tot * (1.f / WS_K) + g.ln_eps);
            float mean_c[2], rstd_c[2];
            lane_swap16(mean, mean_c[0], mean_c[1]);
            lane_swap16(rstd, rstd_c[0], rstd_c[1]);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              bf16_t* xp = g.XN + (row0 + 16 * c) * WS_K + n0 + cbase16;
              uint2 o[2];
#pragma unroll
              for (int b = 0; b < 2; ++b) {
                const int nl = wv * 32 + cbase16 + CSTEP * b;
                if constexpr (AFF) {
                  const float4 gg = *reinterpret_cast<const float4*>(gam_lds + nl), bb = *reinterpret_cast<const float4*>(bet_lds + nl);
                  o[b] = make_uint2(pack2<MODE>((acc[c][b][0] - mean_c[c]) * rstd_c[c] * gg.x + bb.x, (acc[c][b][1] - mean_c[c]) * rstd_c[c] * gg.y + bb.y),
                                    pack2<MODE>((acc[c][b][2] - mean_c[c]) * rstd_c[c] * gg.z + bb.z, (acc[c][b][3] - mean_c[c]) * rstd_c[c] * gg.w + bb.w));
                } else {
                  o[b] = make_uint2(pack2<MODE>((acc[c][b][0] - mean_c[c]) * rstd_c[c], (acc[c][b][1] - mean_c[c]) * rstd_c[c]),
                                    pack2<MODE>((acc[c][b][2] - mean_c[c]) * rstd_c[c], (acc[c][b][3] - mean_c[c]) * rstd_c[c]));
                }
              }
              if constexpr (F32OUT) {
                *reinterpret_cast<uint2*>(xp) = o[0];
                *reinterpret_cast<uint2*>(xp + CSTEP) = o[1];
              } else {
                *reinterpret_cast<uint4*>(xp) = make_uint4(o[0].x, o[0].y, o[1].x, o[1].y);
              }
            }
          }
        }
        }   // skip_epi16 (measurement builds)
      } else {
      // ================= 32x32x16 form (rounds 2-3) =================
      const char* pb = panels_lds + buf * WS_PANEL_BYTES + h2 * (32 * WS_K * 2) + frag_r;
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bq = *reinterpret_cast<const float4*>(bias_lds + wv * 32 + 16 * (q >> 1) + 8 * half + 4 * (q & 1));
        acc[4 * q + 0] = bq.x; acc[4 * q + 1] = bq.y; acc[4 * q + 2] = bq.z; acc[4 * q + 3] = bq.w;
      }
      if (EPI == 3 && vslice) {   // exchanged operands: this lane's output column is weight row n0 + pi(r32), all 16 registers
        int to = tid;
        asm volatile("" : "+v"(to));
        const int r32o = to & 31;
        const float bl = bias_lds[(to >> 6) * 32 + 16 * (r32o >> 4) + 8 * ((r32o >> 2) & 1) + 4 * ((r32o >> 3) & 1) + (r32o & 3)];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = bl;
      }
      auto rd = [&](int kb) __attribute__((always_inline)) {
        // The GELU epilogue makes fc1 vector-ISSUE bound (PMC: 8.5 VALU instructions per MFMA; a 32x32x16 MFMA leaves room for
        // six): there the eight fragment addresses are precomputed (registers paid for with one step less of read-ahead).  The
        // other variants have no registers left and recompute the address per read (one v_xad).
        int fx = frag_x;
        asm volatile("" : "+v"(fx));
        return *reinterpret_cast<const bf16x8*>(pb + (((kb & 7) << 5) ^ fx) + (kb >> 3) * 256);
      };
      // fragment reads kept in flight ahead of the MFMA that consumes them (the residual epilogues are HBM-bound and short of
      // registers: one)
      constexpr int WS_DEPTH = (EPI == 2 || EPI == 3) ? 1 : (LN_IN || EPI == 1 || EPI == 4) ? 2 : 3;
      auto mfma_loop = [&](auto swapped_c) __attribute__((always_inline)) {
        constexpr bool SWAPPED = decltype(swapped_c)::value;   // C^T = X W^T: lane = output column, registers = rows (EPI 3, v slice)
        bf16x8 f[WS_DEPTH + 1];
#pragma unroll
        for (int kb = 0; kb < WS_DEPTH; ++kb) f[kb] = rd(kb);
#pragma unroll
        for (int kb = 0; kb < 24; ++kb) {
          if (kb + WS_DEPTH < 24 && !(abl & 8)) f[(kb + WS_DEPTH) % (WS_DEPTH + 1)] = rd(kb + WS_DEPTH);
          __builtin_amdgcn_sched_barrier(0);
          if (!(abl & 4)) {
            if constexpr (SWAPPED) Mma32<MODE>::mma(acc, f[kb % (WS_DEPTH + 1)], w[kb]);
            else Mma32<MODE>::mma(acc, w[kb], f[kb % (WS_DEPTH + 1)]);
          } else if (kb == 23) {   // measurement builds: keep every fragment register live (WS_DEPTH + 1 of them, not a fixed four)
#pragma unroll
            for (int i = 0; i <= WS_DEPTH; ++i) acc[0] += (float)f[i][i];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      if constexpr (EPI == 3) {
        if (vslice) mfma_loop(std::true_type{});
        else mfma_loop(std::false_type{});
      } else {
        mfma_loop(std::false_type{});
      }

      // ---- epilogue: lane (r32, half) holds, of row m0 + 32 h2 + r32, the columns n0 + 16 qp + 8 half + (0..7), qp = 0, 1
      const int64_t row = m0 + 32 * h2 + r32;
      bool skip_epi = false;
      if constexpr ((abl & 1) != 0) skip_epi = acc[3] != 1234.5f;
      if (!skip_epi) {
      if constexpr (EPI == 3) {
        float amax = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[e] *= scale; amax = fmaxf(amax, fabsf(acc[e])); }
        float ma, mb;
        lane_swap32(amax, ma, mb);
        float inv;
        const unsigned eb = mx_scale_byte(fmaxf(ma, mb), inv);
        unsigned qd[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) qd[q4] = mx_cvt4(acc[4 * q4] * inv, acc[4 * q4 + 1] * inv, acc[4 * q4 + 2] * inv, acc[4 * q4 + 3] * inv);
        // every per-lane address below is derived from an opaque copy of the thread index: the compiler then cannot hoist them
        // out of the panel loop into permanently live registers (this kernel has none to spare)
        int to = tid;
        asm volatile("" : "+v"(to));
        const int r32o = to & 31, halfo = (to >> 5) & 1, wvo = to >> 6;
        if (!vslice) {
          // lane (row r32, half): columns 16 qp + 8 half + (0..7) of the wave's 32 = dwords (2 qp, 2 qp + 1); scale plane = wv
          const int64_t rowo = m0 + 32 * h2 + r32o;
          unsigned char* dst = (slice == 0 ? g.mx.q8 : g.mx.k8) + rowo * MX_DIM + wvo * 32 + 8 * halfo;
          *reinterpret_cast<uint2*>(dst) = make_uint2(qd[0], qd[1]);
          *reinterpret_cast<uint2*>(dst + 16) = make_uint2(qd[2], qd[3]);
          if (halfo == 0) (slice == 0 ? g.mx.sq : g.mx.sk)[(int64_t)wvo * g.mx.rows_alloc + rowo] = (unsigned char)eb;
        } else {
          // lane (d column, half): registers (g, i) = token 8 g + 4 half + i of the half panel.  Two lane swaps hand each half 16
          // CONSECUTIVE tokens: half 0 -> (own q0, other q0, own q1, other q1) = tokens 0..15, half 1 -> tokens 16..31
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(qd[0]), "+v"(qd[2]));
          asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(qd[1]), "+v"(qd[3]));
          const int dcol = wvo * 32 + 16 * (r32o >> 4) + 8 * ((r32o >> 2) & 1) + 4 * ((r32o >> 3) & 1) + (r32o & 3);
          const int64_t tok0 = m0 + 32 * h2;
          *reinterpret_cast<uint4*>(g.mx.v8t + (int64_t)dcol * g.mx.rows_alloc + tok0 + 16 * halfo) = make_uint4(qd[0], qd[2], qd[1], qd[3]);
          if (halfo == 0) g.mx.sv[(tok0 >> 5) * MX_DIM + dcol] = (unsigned char)eb;
        }
      } else if constexpr (EPI != 2) {
        bf16_t* cp = reinterpret_cast<bf16_t*>(g.C) + row * g.ldc + n0 + 8 * half;
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          v2f v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v2f{acc[8 * qp + 2 * e], acc[8 * qp + 2 * e + 1]};
          if constexpr (EPI == 4) {
            *reinterpret_cast<uint4*>(cp + 16 * qp) = make_uint4(pg_gelu_h2(v[0]), pg_gelu_h2(v[1]), pg_gelu_h2(v[2]), pg_gelu_h2(v[3]));
            continue;
          } else if constexpr (EPI == 1) {
            pg_gelu4(v[0], v[1]);
            pg_gelu4(v[2], v[3]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e].x *= scale; v[e].y *= scale; }
          }
          *reinterpret_cast<uint4*>(cp + 16 * qp) = make_uint4(pack2<MODE>(v[0].x, v[0].y), pack2<MODE>(v[1].x, v[1].y),
                                                                 pack2<MODE>(v[2].x, v[2].y), pack2<MODE>(v[3].x, v[3].y));
        }
      } else {
        float* cp = reinterpret_cast<float*>(g.C) + row * g.ldc + n0 + 8 * half;
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {     // two 16-byte pieces at a time: the residual epilogues are short of registers
          const float4 x0 = *reinterpret_cast<const float4*>(cp + 16 * qp), x1 = *reinterpret_cast<const float4*>(cp + 16 * qp + 4);
          acc[8 * qp + 0] += x0.x; acc[8 * qp + 1] += x0.y; acc[8 * qp + 2] += x0.z; acc[8 * qp + 3] += x0.w;
          acc[8 * qp + 4] += x1.x; acc[8 * qp + 5] += x1.y; acc[8 * qp + 6] += x1.z; acc[8 * qp + 7] += x1.w;
          *reinterpret_cast<float4*>(cp + 16 * qp) = make_float4(acc[8 * qp], acc[8 * qp + 1], acc[8 * qp + 2], acc[8 * qp + 3]);
          *reinterpret_cast<float4*>(cp + 16 * qp + 4) = make_float4(acc[8 * qp + 4], acc[8 * qp + 5], acc[8 * qp + 6], acc[8 * qp + 7]);
        }
        if constexpr (LN_OUT) {
          // LayerNorm of the updated rows: per wave (mean, M2) of its 32 columns of a row (two passes over registers, the two
          // lane halves combined by a lane swap), the 12 partials of a row merged with the parallel-variance formula.
          float s = 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) s += acc[e];
          float sa, sb;
          lane_swap32(s, sa, sb);
          const float mu = (sa + sb) * (1.f / 32.f);
          float m2 = 0.f;
#pragma unroll
          for (int e = 0; e < 16; ++e) { const float d = acc[e] - mu; m2 += d * d; }
          lane_swap32(m2, sa, sb);
          // [wave][row]: the 32 rows of a lane group are consecutive 8-byte slots (conflict-free).  The first layout, [row][wave]
          // with a 24-dword row stride, put rows r and r + 4 on the same banks: every one of these reads was an 8-way conflict,
          // 48 % of the kernel's LDS cycles (profiles/r3_e_kernel_pmc.json) -- twice the panel's MFMA time on the CU's one LDS.
          float2* sp = stat_lds + (32 * h2 + r32);
          if (half == 0) sp[wv * WS_BM] = make_float2(mu, sa + sb);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          float mean = 0.f;
#pragma unroll 4
          for (int k = 0; k < WS_WAVES; ++k) mean += sp[k * WS_BM].x;
          mean *= 1.f / WS_WAVES;
          float tot = 0.f;
#pragma unroll 4
          for (int k = 0; k < WS_WAVES; ++k) { const float2 st = sp[k * WS_BM]; const float d = st.x - mean; tot += st.y + 32.f * d * d; }
          const float rstd = rsqrtf(tot * (1.f / WS_K) + g.ln_eps);
          bf16_t* xp = g.XN + row * WS_K + n0 + 8 * half;
#pragma unroll
          for (int qp = 0; qp < 2; ++qp) {
            const int nl = wv * 32 + 16 * qp + 8 * half;
            unsigned o[4];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              float4 gg = make_float4(1.f, 1.f, 1.f, 1.f), bb = make_float4(0.f, 0.f, 0.f, 0.f);
              if constexpr (AFF) { gg = *reinterpret_cast<const float4*>(gam_lds + nl + 4 * e); bb = *reinterpret_cast<const float4*>(bet_lds + nl + 4 * e); }
              o[2 * e] = pack2<MODE>((acc[8 * qp + 4 * e + 0] - mean) * rstd * gg.x + bb.x, (acc[8 * qp + 4 * e + 1] - mean) * rstd * gg.y + bb.y);
              o[2 * e + 1] = pack2<MODE>((acc[8 * qp + 4 * e + 2] - mean) * rstd * gg.z + bb.z, (acc[8 * qp + 4 * e + 3] - mean) * rstd * gg.w + bb.w);
            }
            *reinterpret_cast<uint4*>(xp + 16 * qp) = make_uint4(o[0], o[1], o[2], o[3]);
          }
        }
      }
      }   // skip_epi (measurement builds)
      }   // SH
      WS_T(2)                                           // [2] epilogue (arithmetic + issue of its stores)
      if constexpr (SPLIT) {
        if (more) {
          WS_WAIT_FETCH(S / 2)         // younger than the fetch: this half-panel's S / 2 epilogue operations
          WS_DEPOSIT_HALF(buf ^ 1, h2)
        }
      }
      if constexpr (LN_IN) {
        if (more) {
          WS_WAIT_FETCH(S / 2)
          WS_T(3)                      // [3] waiting for the fetched panel (+ the acknowledgement of older stores)
          WS_DEPOSIT_LN(buf ^ 1, table_lds + ((it + 1) % 3) * WS_BM, h2)
          if (stats_next && wv == 0) table_lds[((it + 2) % 3) * WS_BM + lane] = merge_stats(st01, st2);
          WS_T(4)                      // [4] deposit (LayerNorm arithmetic + LDS writes)
        }
      }
    }
    if constexpr (!SPLIT && !LN_IN) {
      if (more) {
        WS_WAIT_FETCH(S)               // younger than the fetch: exactly this panel's S epilogue operations
        WS_T(3)
        WS_DEPOSIT(buf ^ 1)            // the buffer panel p-1 left at this iteration's barrier
        WS_T(4)
      }
    }
  }
#ifdef WS_STAMP
  if (lane == 0) {
    unsigned long long* o = ws_stamps + ((size_t)blockIdx.x * WS_WAVES + wv) * 8;
    for (int i = 0; i < 5; ++i) o[i] = st_acc[i];
    o[5] = (unsigned long long)(p1 - p0);
  }
#endif
}

static int ws_gemm_launch(const char* who, const void* A, const float* X, const float* row_stats, const void* W, const float* bias,
                          void* C, int ldc, int64_t M, int N, int epilogue, int qscale_cols, float qscale, void* xn_out,
                          const float* ln_gamma, const float* ln_beta, float ln_eps, int dtype, void* stream, void* mx_ws = nullptr) {
  static int n_cu_dev[64] = {0};
  int dev = 0;
  MAAVSS_CHECK_ARG(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, "%s: cannot query the device", who);
  if (!n_cu_dev[dev]) {
    hipDeviceProp_t prop;
    MAAVSS_CHECK_ARG(hipGetDeviceProperties(&prop, dev) == hipSuccess, "%s: cannot query the device", who);
    n_cu_dev[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int cu_per_xcd = n_cu_dev[dev] / 8 > 0 ? n_cu_dev[dev] / 8 : 1;
  WsArgs g;
  g.A = (const bf16_t*)A; g.X = X; g.row_stats = row_stats; g.W = (const bf16_t*)W; g.bias = bias; g.C = C; g.XN = (bf16_t*)xn_out;
  g.ln_g = ln_gamma; g.ln_b = ln_beta; g.ln_eps = ln_eps;
  g.M = (int)M; g.N = N; g.ldc = ldc; g.qscale_cols = qscale_cols; g.qscale = qscale;
  if (mx_ws) g.mx = mx_images(mx_ws, M);
  g.ns = N / WS_SLICE;
  MAAVSS_CHECK_ARG(g.ns <= cu_per_xcd, "%s: N too large for one XCD's CUs", who);
  g.groups_per_xcd = cu_per_xcd / g.ns;
  g.panels = cdiv(M, WS_BM);
  const size_t smem = WS_BUFS * WS_PANEL_BYTES + 3 * WS_SLICE * sizeof(float) + WS_BM * WS_WAVES * sizeof(float2);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(8 * cu_per_xcd), block(WS_THREADS);
  // MFMA shape: 16x16x32 unless MAAVSS_WS_SHAPE=32 asks for the round-2/3 form (A/B runs; read once per process); the MX epilogue is 32x32 only
  static int shape_env = 0;
  if (!shape_env) {
    const char* e = getenv("MAAVSS_WS_SHAPE");
    shape_env = (e && e[0] == '3') ? 32 : 16;
  }
#define WS_LAUNCH4(E, L, D, H)                                                                                        \
  {                                                                                                                   \
    hipFuncSetAttribute(reinterpret_cast<const void*>(vit_ws_gemm_kernel<E, L, D, H>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL((vit_ws_gemm_kernel<E, L, D, H>), grid, block, smem, st, g);                                   \
  }
#define WS_LAUNCH3(E, L, D) { if (shape_env == 32) WS_LAUNCH4(E, L, D, 32) else WS_LAUNCH4(E, L, D, 16) }
#define WS_LAUNCH(E, L) { if (dtype == MODE_F16) WS_LAUNCH3(E, L, MODE_F16) else WS_LAUNCH3(E, L, MODE_BF16) }
  const bool affine = ln_gamma != nullptr;          // null gamma / beta: the LayerNorm variants without the affine part (folded into the consumer's weights)
  if (X && epilogue == 3) {
    if (affine) { if (dtype == MODE_F16) WS_LAUNCH4(3, 2, MODE_F16, 32) else WS_LAUNCH4(3, 2, MODE_BF16, 32) }
    else { if (dtype == MODE_F16) WS_LAUNCH4(3, 4, MODE_F16, 32) else WS_LAUNCH4(3, 4, MODE_BF16, 32) }
  }
  else if (X && epilogue == 5) { if (dtype == MODE_F16) WS_LAUNCH4(0, 5, MODE_F16, 16) else WS_LAUNCH4(0, 5, MODE_BF16, 16) }
  else if (X && affine) WS_LAUNCH(0, 2) else if (X) WS_LAUNCH(0, 4) else if (epilogue == 0) WS_LAUNCH(0, 0) else if (epilogue == 1) WS_LAUNCH(1, 0)
  else if (epilogue == 4) WS_LAUNCH3(4, 0, MODE_F16) else if (xn_out && affine) WS_LAUNCH(2, 1) else if (xn_out) WS_LAUNCH(2, 3) else WS_LAUNCH(2, 0)
#undef WS_LAUNCH4
#undef WS_LAUNCH3
#undef WS_LAUNCH
  MAAVSS_LAUNCH_CHECK("vit_ws_gemm_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_vit_ws_gemm(const void* A, int lda, int64_t a_rows, const void* W, const float* bias, void* C, int ldc, int64_t c_rows,
                                  int64_t M, int N, int epilogue, int qscale_cols, float qscale, void* xn_out,
                                  const float* ln_gamma, const float* ln_beta, float ln_eps, int dtype, void* stream) {
  MAAVSS_CHECK_ARG(A && W && bias && C && M > 0 && M < (1LL << 31), "vit_ws_gemm: bad arguments");
  MAAVSS_CHECK_ARG(N % WS_SLICE == 0 && N >= WS_SLICE, "vit_ws_gemm: N must be a multiple of 384 (got %d)", N);
  MAAVSS_CHECK_ARG((epilogue >= 0 && epilogue <= 2) || epilogue == 4, "vit_ws_gemm: unknown epilogue");
  MAAVSS_CHECK_ARG(epilogue != 4 || dtype == MODE_F16, "vit_ws_gemm: epilogue 4 (GELU evaluated in packed half) needs the IEEE-half storage format (dtype 2)");
  MAAVSS_CHECK_ARG(dtype == MODE_BF16 || dtype == MODE_F16, "vit_ws_gemm: dtype must be 0 (bf16) or 2 (f16)");
  MAAVSS_CHECK_ARG(lda == WS_K, "vit_ws_gemm: A must be dense [rows][384] (lda = %d)", lda);
  MAAVSS_CHECK_ARG(a_rows >= (int64_t)cdiv(M, WS_BM) * WS_BM, "vit_ws_gemm: A needs ceil(M/64)*64 = %ld allocated rows (got %ld): whole panels are read",
                   (long)cdiv(M, WS_BM) * WS_BM, (long)a_rows);
  MAAVSS_CHECK_ARG(ldc % 8 == 0 && ldc >= N && qscale_cols % WS_SLICE == 0, "vit_ws_gemm: ldc must be a multiple of 8, qscale_cols a multiple of 384");
  MAAVSS_CHECK_ARG(c_rows >= (int64_t)cdiv(M, WS_BM) * WS_BM, "vit_ws_gemm: C needs ceil(M/64)*64 = %ld allocated rows (got %ld): stores are unguarded",
                   (long)cdiv(M, WS_BM) * WS_BM, (long)c_rows);
  MAAVSS_CHECK_ARG(!xn_out || (epilogue == 2 && N == WS_SLICE && ((ln_gamma != nullptr) == (ln_beta != nullptr))),
                   "vit_ws_gemm: the LayerNorm output needs epilogue 2, N = 384 and ln_gamma / ln_beta both given or both null (null = no affine part)");
  return ws_gemm_launch("vit_ws_gemm", A, nullptr, nullptr, W, bias, C, ldc, M, N, epilogue, qscale_cols, qscale, xn_out, ln_gamma, ln_beta,
                        ln_eps, dtype, stream);
}

extern "C" int maavss_vit_ws_gemm_ln(const float* X, int64_t x_rows, const float* row_stats, const float* ln_gamma, const float* ln_beta,
                                     float ln_eps, const void* W, const float* bias, void* C, int ldc, int64_t c_rows, int64_t M, int N,
                                     int qscale_cols, float qscale, int dtype, void* stream) {
  MAAVSS_CHECK_ARG(X && row_stats && W && bias && C && M > 0 && M < (1LL << 31), "vit_ws_gemm_ln: bad arguments");
  MAAVSS_CHECK_ARG((ln_gamma != nullptr) == (ln_beta != nullptr), "vit_ws_gemm_ln: ln_gamma / ln_beta both given or both null (null = no affine part)");
  MAAVSS_CHECK_ARG(N % WS_SLICE == 0 && N >= WS_SLICE, "vit_ws_gemm_ln: N must be a multiple of 384 (got %d)", N);
  MAAVSS_CHECK_ARG(dtype == MODE_BF16 || dtype == MODE_F16, "vit_ws_gemm_ln: dtype must be 0 (bf16) or 2 (f16)");
  MAAVSS_CHECK_ARG(x_rows >= (int64_t)cdiv(M, WS_BM) * WS_BM && c_rows >= (int64_t)cdiv(M, WS_BM) * WS_BM,
                   "vit_ws_gemm_ln: X and C need ceil(M/64)*64 = %ld allocated rows (got %ld, %ld): whole panels are read and stored",
                   (long)cdiv(M, WS_BM) * WS_BM, (long)x_rows, (long)c_rows);
  MAAVSS_CHECK_ARG(ldc % 8 == 0 && ldc >= N && qscale_cols % WS_SLICE == 0, "vit_ws_gemm_ln: ldc must be a multiple of 8, qscale_cols a multiple of 384");
  return ws_gemm_launch("vit_ws_gemm_ln", nullptr, X, row_stats, W, bias, C, ldc, M, N, 0, qscale_cols, qscale, nullptr, ln_gamma, ln_beta, ln_eps,
                        dtype, stream);
}

// attn.qkv with the LayerNorm applied AFTER the product (kernel variant LN = 5): W = W diag(gamma) in the storage format, col_sums[n] = sum_k of
// that (rounded) W's row n, bias = b + W beta.  Same result as maavss_vit_ws_gemm_ln in exact arithmetic, another rounding realisation in 16 bits
// (the raw rows are rounded instead of the normalised ones).
extern "C" int maavss_vit_ws_gemm_ln_post(const float* X, int64_t x_rows, const float* row_stats, const float* col_sums, float ln_eps, const void* W,
                                          const float* bias, void* C, int ldc, int64_t c_rows, int64_t M, int N, int qscale_cols, float qscale,
                                          int dtype, void* stream) {
  MAAVSS_CHECK_ARG(X && row_stats && col_sums && W && bias && C && M > 0 && M < (1LL << 31), "vit_ws_gemm_ln_post: bad arguments");
  MAAVSS_CHECK_ARG(N % WS_SLICE == 0 && N >= WS_SLICE, "vit_ws_gemm_ln_post: N must be a multiple of 384 (got %d)", N);
  MAAVSS_CHECK_ARG(dtype == MODE_BF16 || dtype == MODE_F16, "vit_ws_gemm_ln_post: dtype must be 0 (bf16) or 2 (f16)");
  MAAVSS_CHECK_ARG(x_rows >= (int64_t)cdiv(M, WS_BM) * WS_BM && c_rows >= (int64_t)cdiv(M, WS_BM) * WS_BM,
                   "vit_ws_gemm_ln_post: X and C need ceil(M/64)*64 = %ld allocated rows (got %ld, %ld): whole panels are read and stored",
                   (long)cdiv(M, WS_BM) * WS_BM, (long)x_rows, (long)c_rows);
  MAAVSS_CHECK_ARG(ldc % 8 == 0 && ldc >= N && qscale_cols % WS_SLICE == 0, "vit_ws_gemm_ln_post: ldc must be a multiple of 8, qscale_cols a multiple of 384");
  return ws_gemm_launch("vit_ws_gemm_ln_post", nullptr, X, row_stats, W, bias, C, ldc, M, N, 5, qscale_cols, qscale, nullptr, col_sums, nullptr, ln_eps,
                        dtype, stream);
}

// attn.qkv with LayerNorm on the way in (as maavss_vit_ws_gemm_ln) writing the block-scaled fp8 operand images of
// maavss_vit_attn_mx into `mx_ws` (maavss_vit_attn_mx_ws_bytes(M) bytes) instead of a 16-bit qkv tensor.  N = 1152.
extern "C" int maavss_vit_ws_gemm_ln_mx(const float* X, int64_t x_rows, const float* row_stats, const float* ln_gamma, const float* ln_beta,
                                        float ln_eps, const void* W, const float* bias, void* mx_ws, int64_t M, int qscale_cols, float qscale,
                                        int dtype, void* stream) {
  MAAVSS_CHECK_ARG(X && row_stats && W && bias && mx_ws && M > 0 && M < (1LL << 31), "vit_ws_gemm_ln_mx: bad arguments");
  MAAVSS_CHECK_ARG((ln_gamma != nullptr) == (ln_beta != nullptr), "vit_ws_gemm_ln_mx: ln_gamma / ln_beta both given or both null (null = no affine part)");
  MAAVSS_CHECK_ARG(dtype == MODE_BF16 || dtype == MODE_F16, "vit_ws_gemm_ln_mx: dtype (of the weights) must be 0 (bf16) or 2 (f16)");
  MAAVSS_CHECK_ARG(x_rows >= (int64_t)cdiv(M, WS_BM) * WS_BM, "vit_ws_gemm_ln_mx: X needs ceil(M/64)*64 = %ld allocated rows (got %ld): whole panels are read",
                   (long)cdiv(M, WS_BM) * WS_BM, (long)x_rows);
  MAAVSS_CHECK_ARG(qscale_cols % WS_SLICE == 0 && ((uintptr_t)mx_ws & 255) == 0, "vit_ws_gemm_ln_mx: qscale_cols must be a multiple of 384, mx_ws 256-byte aligned");
  MAAVSS_CHECK_ARG(mx_ws_bytes(M) < (1LL << 32), "vit_ws_gemm_ln_mx: %ld rows: workspace beyond the attention kernel's 32-bit offsets (launch fewer frames per group)", (long)M);
  return ws_gemm_launch("vit_ws_gemm_ln_mx", nullptr, X, row_stats, W, bias, mx_ws, 3 * WS_SLICE, M, 3 * WS_SLICE, 3, qscale_cols, qscale, nullptr,
                        ln_gamma, ln_beta, ln_eps, dtype, stream, mx_ws);
}
