// K2+K3 fused: the K = 384 dense layers of the ViT blocks (qkv, proj, mlp.fc1) with the preceding LayerNorm folded
// into the GEMM -- dino Block: norm1 -> attn.qkv, attn.proj, norm2 -> mlp.fc1 (reached from the reference through
// video_attention.py:52).  Why a second GEMM kernel: at K = 384 the tiled kernel (vit_gemm.hip) is bound by the
// L2 -> LDS feed (48 KiB of operands per 256x128x64 step) and by epilogues that nothing overlaps.  Here the
// activation panel is STATIONARY:
//   * a workgroup owns 128 rows; the whole [128 x 384] bf16 panel lives in LDS (96 KiB) and is built once -- either
//     by LayerNorm-ing the f32 residual rows on the way in (one wave per row, wave reductions; the normalised tensor
//     never exists in HBM and the LayerNorm kernel disappears), or by copying a bf16 input;
//   * only the weights stream (128 FLOP per staged byte): [128 n x 32 k] tiles (8 KiB, L2-resident) through an
//     8-slot global_load_lds ring that fills the remaining 64 KiB of LDS and runs continuously over all N tiles of
//     the panel.  The ring is issued by a dedicated ninth wave and kept SIX steps (48 KiB) ahead: the L2 -> LDS
//     path answers after ~1300 clocks under load (s_memtime stamps), so the bytes in flight, not the MFMA rate,
//     set the speed of this loop;
//   * the MFMA roles are swapped (weights = A operand, activations = B operand), so an accumulator lane holds
//     4 CONSECUTIVE output columns of one row: the epilogue stores 16 B straight from registers, deferred by one
//     N tile and sliced under the next tile's MFMAs.
// Epilogues: 0 +bias, q-scale -> bf16 | 1 +bias, GELU -> bf16 | 2 +bias +residual -> f32 in place.
#include "mma.h"
#include "vit_epilogue.h"

#define PG_K 384
#define PG_BM 128
#define PG_BN 128
#define PG_BK 32
#define PG_STAGES 8
#define PG_MMA_WAVES 8
#define PG_LOADERS 4                       // weight-stream (LDS-DMA) waves, one per SIMD
#define PG_THREADS ((PG_MMA_WAVES + PG_LOADERS) * 64)
#define PG_PANEL_ELEMS (PG_BM * PG_K)      // 96 KiB of bf16
#define PG_BTILE_ELEMS (PG_BN * PG_BK)     // 8 KiB of bf16
#define PG_NKS (PG_K / PG_BK)              // 12 K-steps per N tile

struct PGemmArgs {
  const float* X;        // f32 [M][384] (LayerNorm fused) or null
  const bf16_t* A;       // bf16 [M][lda] when X is null
  const float* ln_g;
  const float* ln_b;
  float ln_eps;
  const bf16_t* W;       // [N][384]
  const float* bias;     // [N]
  void* C;               // bf16 [M][ldc] (epi 0,1) or f32 [M][ldc] (epi 2)
  int M, N, lda, ldc;
  int qscale_cols;
  float qscale;
  int panels;
  int full, split;   // workgroups [0, full): whole panels; the rest: panels of the tail round, `split` workgroups each
};

// element offset of (row r, k) inside the LDS panel: 6 segments of 64 k (128 B) per row, 16-B chunk ^= r & 7
__device__ __forceinline__ int panel_off(int r, int k) {
  return r * PG_K + (k & ~63) + ((((k & 63) >> 3) ^ (r & 7)) << 3) + (k & 7);
}

// LayerNorm the NP row PAIRS [r0, r0 + 2 NP) of the panel that starts at global row m0, one wave.  Two consecutive rows
// are 768 contiguous floats = three 16-byte-per-lane loads (j = 0: row A [4l, +4); j = 1: lanes < 32 row A [256 + 4l, +4),
// lanes >= 32 row B [4(l-32), +4); j = 2: row B [128 + 4l, +4)): half the vector-memory instructions of an 8-byte-per-lane
// sweep (the issue of 42 such loads per wave, twelve waves at once, was stamped at 8-13 k cycles per panel).  All 3 NP
// loads go out before the first reduction, so the wave pays ONE memory latency.
template <int NP, int MODE>
__device__ __forceinline__ void pg_ln_rows(const PGemmArgs& g, bf16_t* panel, int m0, int r0, int lane) {
  const bool lo = lane < 32;
  const int kj[3] = {4 * lane, lo ? 256 + 4 * lane : 4 * (lane - 32), 128 + 4 * lane};
  float4 gam[3], bet[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    gam[j] = *reinterpret_cast<const float4*>(g.ln_g + kj[j]);
    bet[j] = *reinterpret_cast<const float4*>(g.ln_b + kj[j]);
  }
  float4 v[NP][3];
#pragma unroll
  for (int pp = 0; pp < NP; ++pp) {
    const int gr = m0 + r0 + 2 * pp;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      int row = gr + (j == 0 ? 0 : j == 2 ? 1 : (lo ? 0 : 1));
      row = row < g.M ? row : g.M - 1;   // rows past the end repeat the last one (their outputs land in C's padding rows)
      v[pp][j] = *reinterpret_cast<const float4*>(g.X + (int64_t)row * PG_K + kj[j]);
    }
  }
#pragma unroll
  for (int pp = 0; pp < NP; ++pp) {
    auto sum4 = [](const float4& a) __attribute__((always_inline)) { return (a.x + a.y) + (a.z + a.w); };
    const float s0 = sum4(v[pp][0]), s1 = sum4(v[pp][1]), s2 = sum4(v[pp][2]);
    const float meanA = wave_sum(s0 + (lo ? s1 : 0.f)) * (1.f / PG_K);
    const float meanB = wave_sum(s2 + (lo ? 0.f : s1)) * (1.f / PG_K);
    const float mean[3] = {meanA, lo ? meanA : meanB, meanB};
    float4 c[3];
    float q[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      c[j] = make_float4(v[pp][j].x - mean[j], v[pp][j].y - mean[j], v[pp][j].z - mean[j], v[pp][j].w - mean[j]);
      q[j] = (c[j].x * c[j].x + c[j].y * c[j].y) + (c[j].z * c[j].z + c[j].w * c[j].w);
    }
    const float rstdA = rsqrtf(wave_sum(q[0] + (lo ? q[1] : 0.f)) * (1.f / PG_K) + g.ln_eps);
    const float rstdB = rsqrtf(wave_sum(q[2] + (lo ? 0.f : q[1])) * (1.f / PG_K) + g.ln_eps);
    const float rstd[3] = {rstdA, lo ? rstdA : rstdB, rstdB};
    const int row[3] = {r0 + 2 * pp, r0 + 2 * pp + (lo ? 0 : 1), r0 + 2 * pp + 1};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const uint2 o = make_uint2(pack2<MODE>(c[j].x * rstd[j] * gam[j].x + bet[j].x, c[j].y * rstd[j] * gam[j].y + bet[j].y),
                                 pack2<MODE>(c[j].z * rstd[j] * gam[j].z + bet[j].z, c[j].w * rstd[j] * gam[j].w + bet[j].w));
      *reinterpret_cast<uint2*>(panel + panel_off(row[j], kj[j])) = o;
    }
  }
}

template <int EPI, bool FUSE_LN, int MODE>
__global__ __launch_bounds__(PG_THREADS) void vit_panel_gemm_kernel(PGemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* panel = reinterpret_cast<bf16_t*>(smem);
  bf16_t* ring = panel + PG_PANEL_ELEMS;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l16 = lane & 15, gq = lane >> 4;
  const int wn = (wv >> 2) & 1, wm = wv & 3;   // wave tile: 64 output columns (n) x 32 rows (m)
  // Work split.  Workgroups [0, full) take one panel and all N tiles each.  The panels of the last, partly filled round
  // (panels mod #CUs of them: 68 of 3140 on 256 CUs, a round that would leave 188 CUs idle for a whole panel time) are
  // split over `split` workgroups that each build the panel and take a contiguous share of its N tiles.
  int pidx = blockIdx.x, tn0 = 0, ntiles = g.N / PG_BN;
  if ((int)blockIdx.x >= g.full) {
    const int idx = (int)blockIdx.x - g.full, part = idx % g.split;
    pidx = g.full + idx / g.split;
    tn0 = part * ntiles / g.split;
    ntiles = (part + 1) * ntiles / g.split - tn0;
  }
  const int m0 = pidx * PG_BM;
  const int total_steps = ntiles * PG_NKS;

  // ---- weight stream: four dedicated loader waves (one per SIMD), two 1-KiB pieces each per step.  Dedicated
  // because (1) vmcnt retires in issue order -- a wave that both stores outputs and waits for DMA loads waits for the
  // acknowledgement of every older store first -- and (2) one global_load_lds wave-instruction costs its wave ~60
  // issue cycles (stamped: a single loader wave tops out at 17 B/clk, half of what the MFMAs consume), cycles that
  // must not come out of the MFMA waves' instruction streams.  The loaders wait for their own loads only and publish
  // them through the per-step barrier.  One step = [128 n x 32 k] = 8 pieces of 16 rows x 64 B.  LDS row r (64 B =
  // four 16-B chunks) holds source chunk c at position c ^ ((r >> 2) & 3): the swizzle is applied on the global side
  // (the LDS side of the DMA is lane-linear) and makes the ds_read_b128 fragment reads conflict-free.
  const int srow = lane >> 2, schunk = lane & 3;
  const int lw = wv - PG_MMA_WAVES;   // loader index (negative on MFMA waves)
  const bf16_t* wsrc = g.W + (int64_t)(srow + lw * 32) * PG_K + ((schunk ^ ((srow >> 2) & 3)) * 8);
  auto stage = [&](int u) {   // virtual step u = tile * 12 + kstep -> ring slot u % 8; this loader's rows [32 lw, +32)
    const int tl = u / PG_NKS, ks = u - tl * PG_NKS, tn = tn0 + tl;
    bf16_t* lb = ring + (u % PG_STAGES) * PG_BTILE_ELEMS + lw * 32 * PG_BK + lane * 8;
    const bf16_t* src = wsrc + (int64_t)tn * PG_BN * PG_K + ks * PG_BK;
#pragma unroll
    for (int i = 0; i < 2; ++i) vit_glds16(src + i * 16 * PG_K, lb + i * 16 * PG_BK);
  };
  if (wv >= PG_MMA_WAVES) {
    // Publication and slot release work on PAIRS of steps (one barrier per 64 k): pair P = steps 2P, 2P+1 lives in
    // ring slots (2P, 2P+1) mod 8.  Pairs 0..2 go out at once; iteration P then issues pair P+3 into the slots freed
    // by barrier P-1 (pair P-1's fragment reads retired before it) and waits until pair P+1 has landed: vmcnt counts
    // the 4 instructions of each of the 2 younger pairs.
    const int total_pairs = total_steps / 2;
#pragma unroll
    for (int pp = 0; pp < 3; ++pp)
      if (pp < total_pairs) { stage(2 * pp); stage(2 * pp + 1); }
    if constexpr (FUSE_LN) {
      pg_ln_rows<2, MODE>(g, panel, m0, 112 + lw * 4, lane);     // its loads retire behind the DMA pieces: everything has landed after it
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    } else {
      if (total_pairs > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();          // matches the "panel complete" barrier of the MFMA waves; pair 0 is in LDS
    for (int pp = 0; pp + 1 < total_pairs; ++pp) {
      if (pp + 3 < total_pairs) {
        stage(2 * pp + 6);
        stage(2 * pp + 7);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tail: nothing new goes out, drain
      }
      __builtin_amdgcn_s_barrier();
    }
    return;
  }
  static_assert(PG_STAGES == 8 && PG_NKS % 2 == 0, "pair-granular ring: 4 pairs of slots");

  // ---- build the activation panel
  if constexpr (FUSE_LN) {
    // all twelve waves build the panel, each in a single round of loads: 14 rows per MFMA wave, 4 per loader wave
    // (the loaders first have their 12 ring pieces to issue; stamped, this split lets all waves arrive together)
    pg_ln_rows<7, MODE>(g, panel, m0, wv * 14, lane);
  } else {
    // bf16 input: 8 rows x 128 B per wave-instruction; 16 row-groups x 6 segments = 96 pieces, 12 per wave.  Rows are
    // 768 B apart in the panel, so a lane-linear LDS-DMA destination cannot cover them: staged through registers
    // (16 B per lane), still full-line loads.
    const int prow = lane >> 3, pslot = lane & 7;
#pragma unroll
    for (int p = 0; p < 12; ++p) {
      const int piece = wv * 12 + p, rg = piece / 6, seg = piece % 6;
      const int r = rg * 8 + prow;
      int gr = m0 + r;
      gr = gr < g.M ? gr : g.M - 1;
      const uint4 val = *reinterpret_cast<const uint4*>(g.A + (int64_t)gr * g.lda + seg * 64 + ((pslot ^ (r & 7)) * 8));
      *reinterpret_cast<uint4*>(panel + r * PG_K + seg * 64 + pslot * 8) = val;
    }
  }

  // ---- fragment addressing.  A operand = weights (rows n), B operand = activations (rows m)
  int rown[4], rowm[2];
#pragma unroll
  // MFMA row i of weight tile a is fed with weight row 64 wn + 32 (a>>1) + 8 (i>>2) + 4 (a&1) + (i&3): the D rows a
  // lane owns (i = 4 gq + r) of tiles (2h, 2h+1) are then 8 CONSECUTIVE output columns -> 16-byte stores, and the
  // four lane groups of a row write 64 contiguous bytes.
  for (int a = 0; a < 4; ++a) rown[a] = wn * 64 + (a >> 1) * 32 + (l16 >> 2) * 8 + (a & 1) * 4 + (l16 & 3);
#pragma unroll
  for (int b = 0; b < 2; ++b) rowm[b] = wm * 32 + b * 16 + l16;
  int woff[4], xoff[2];   // per-lane element offsets inside a ring slot / inside a 64-k segment pair of the panel
#pragma unroll
  for (int a = 0; a < 4; ++a) woff[a] = rown[a] * PG_BK + ((gq ^ ((rown[a] >> 2) & 3)) * 8);
#pragma unroll
  for (int b = 0; b < 2; ++b) xoff[b] = rowm[b] * PG_K;
  auto load_frags = [&](int u, bf16x8 (&fw)[4], bf16x8 (&fx)[2]) {
    const int ks = u % PG_NKS;   // 32-k step: 64-k segment ks >> 1, chunks (ks & 1) * 4 + gq of it
    const bf16_t* lb = ring + (u % PG_STAGES) * PG_BTILE_ELEMS;
#pragma unroll
    for (int a = 0; a < 4; ++a) fw[a] = *reinterpret_cast<const bf16x8*>(lb + woff[a]);
#pragma unroll
    for (int b = 0; b < 2; ++b)
      fx[b] = *reinterpret_cast<const bf16x8*>(panel + xoff[b] + (ks >> 1) * 64 + ((((ks & 1) * 4 + gq) ^ (rowm[b] & 7)) * 8));
  };
  f32x4 acc[4][2];
  auto mfmas = [&](const bf16x8 (&fw)[4], const bf16x8 (&fx)[2]) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) Mma<MODE>::mma(acc[a][b], fw[a], fx[b]);
  };
#define PG_USE(fw, fx)               \
  __builtin_amdgcn_sched_barrier(0); \
  asm volatile("" : "+v"(fw[0]), "+v"(fw[1]), "+v"(fw[2]), "+v"(fw[3]), "+v"(fx[0]), "+v"(fx[1]))

  bf16x8 fw0[4], fx0[2], fw1[4], fx1[2];
  __syncthreads();   // panel complete; the DMA wave arrives here with weight steps 0 and 1 and the bias in LDS
  load_frags(0, fw0, fx0);

  // ---- main loop.  The epilogue of N tile tn is DEFERRED: its accumulators are copied to `prev` and written out in
  // four slices during K-steps 0, 2, 4, 6 of tile tn+1, so the bias / GELU / convert VALU work and the stores run underneath
  // this wave's (and its SIMD partner's) MFMAs instead of in a serialized phase of their own.
  f32x4 prev[4][2];
  int prev_tn = -1;
  // slice idx in 0..3 -> (h = idx >> 1, b = idx & 1): the 8 columns n = 128 tn + 64 wn + 32 h + 8 gq + (0..7) of row m
  v2f pbias[2][4];   // bias of the deferred tile: [h][pair] = columns n0 + 64 wn + 32 h + 8 gq + (0..7); no LDS left for it
  auto load_bias = [&](int tn) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float4* bp = reinterpret_cast<const float4*>(g.bias + tn * PG_BN + wn * 64 + h * 32 + gq * 8);
      const float4 b0 = bp[0], b1 = bp[1];
      pbias[h][0] = v2f{b0.x, b0.y}; pbias[h][1] = v2f{b0.z, b0.w}; pbias[h][2] = v2f{b1.x, b1.y}; pbias[h][3] = v2f{b1.z, b1.w};
    }
  };
  auto epi_slice = [&](int idx, int tn) {
    const int b = idx >> 1, h = idx & 1;   // the two 64-byte halves of a row's 128-byte line are written back to back
    const int n = tn * PG_BN + wn * 64 + h * 32 + gq * 8;
    const int m = m0 + rowm[b];
    v2f v[4] = {v2f{prev[2 * h][b][0], prev[2 * h][b][1]} + pbias[h][0], v2f{prev[2 * h][b][2], prev[2 * h][b][3]} + pbias[h][1],
                v2f{prev[2 * h + 1][b][0], prev[2 * h + 1][b][1]} + pbias[h][2], v2f{prev[2 * h + 1][b][2], prev[2 * h + 1][b][3]} + pbias[h][3]};
    // No row guard: C is allocated with ceil(M/128)*128 rows (checked by the launcher).
    if constexpr (EPI == 2) {
      float4* cp = reinterpret_cast<float4*>(reinterpret_cast<float*>(g.C) + (int64_t)m * g.ldc + n);
      const float4 r0 = cp[0], r1 = cp[1];
      cp[0] = make_float4(r0.x + v[0].x, r0.y + v[0].y, r0.z + v[1].x, r0.w + v[1].y);
      cp[1] = make_float4(r1.x + v[2].x, r1.y + v[2].y, r1.z + v[3].x, r1.w + v[3].y);
    } else {
      if constexpr (EPI == 1) {
        pg_gelu4(v[0], v[1]);
        pg_gelu4(v[2], v[3]);
      } else if (n < g.qscale_cols) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * g.qscale;
      }
      const uint4 o = make_uint4(pack2<MODE>(v[0].x, v[0].y), pack2<MODE>(v[1].x, v[1].y), pack2<MODE>(v[2].x, v[2].y), pack2<MODE>(v[3].x, v[3].y));
      *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(g.C) + (int64_t)m * g.ldc + n) = o;
    }
  };

  int u = 0;
  for (int tn = tn0; tn < tn0 + ntiles; ++tn) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool have_prev = prev_tn >= 0;
#pragma unroll
    for (int ks = 0; ks < PG_NKS; ks += 2, u += 2) {
      // even step of the pair: fragments in (fw0, fx0); the odd step is already published, its reads go out at once
      PG_USE(fw0, fx0);
      load_frags(u + 1, fw1, fx1);
      __builtin_amdgcn_sched_barrier(0);     // the reads go out BEFORE the MFMA / epilogue block (hipcc sinks them below it otherwise)
      mfmas(fw0, fx0);
      if (have_prev && (ks == 0 || ks == 4)) {
        epi_slice(ks >> 1, prev_tn);
        epi_slice((ks >> 1) + 1, prev_tn);
      }
      // odd step: once its fragments are in registers the pair's slots are free -> barrier, which publishes the next pair
      PG_USE(fw1, fx1);
      if (u + 2 < total_steps) {
        __builtin_amdgcn_s_barrier();
        load_frags(u + 2, fw0, fx0);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfmas(fw1, fx1);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) prev[a][b] = acc[a][b];
    prev_tn = tn;
    load_bias(tn);   // L2-resident; first used a K-step later
  }
  // flush the last tile's epilogue
#pragma unroll
  for (int idx = 0; idx < 4; ++idx) epi_slice(idx, prev_tn);
}

extern "C" int maavss_vit_panel_gemm(const float* X, const void* A, int lda, const float* ln_gamma, const float* ln_beta,
                                     float ln_eps, const void* W, const float* bias, void* C, int ldc, int64_t c_rows,
                                     int64_t M, int N, int epilogue, int qscale_cols, float qscale, int dtype, void* stream) {
  MAAVSS_CHECK_ARG((X != nullptr) != (A != nullptr), "vit_panel_gemm: exactly one of X (f32, LayerNorm fused) and A (bf16) must be given");
  MAAVSS_CHECK_ARG(W && bias && C && M > 0 && M < (1LL << 31), "vit_panel_gemm: bad arguments");
  MAAVSS_CHECK_ARG(N % PG_BN == 0 && N >= PG_BN, "vit_panel_gemm: N must be a multiple of 128 (got %d)", N);
  MAAVSS_CHECK_ARG(epilogue >= 0 && epilogue <= 2, "vit_panel_gemm: unknown epilogue");
  MAAVSS_CHECK_ARG(dtype == MODE_BF16 || dtype == MODE_F16, "vit_panel_gemm: dtype must be 0 (bf16) or 2 (f16)");
  MAAVSS_CHECK_ARG(!X || (ln_gamma && ln_beta), "vit_panel_gemm: LayerNorm parameters missing");
  MAAVSS_CHECK_ARG(X || (lda % 8 == 0 && lda >= PG_K), "vit_panel_gemm: lda must be a multiple of 8 and >= 384");
  MAAVSS_CHECK_ARG(ldc % 8 == 0 && qscale_cols % 8 == 0, "vit_panel_gemm: ldc / qscale_cols must be multiples of 8");
  MAAVSS_CHECK_ARG(c_rows >= (int64_t)cdiv(M, 128) * 128, "vit_panel_gemm: C needs ceil(M/128)*128 = %ld allocated rows (got %ld): stores are unguarded",
                   (long)cdiv(M, 128) * 128, (long)c_rows);
  PGemmArgs g;
  g.X = X; g.A = (const bf16_t*)A; g.ln_g = ln_gamma; g.ln_b = ln_beta; g.ln_eps = ln_eps;
  g.W = (const bf16_t*)W; g.bias = bias; g.C = C; g.M = (int)M; g.N = N; g.lda = lda; g.ldc = ldc;
  g.qscale_cols = qscale_cols; g.qscale = qscale; g.panels = cdiv(M, PG_BM);
  const size_t smem = (PG_PANEL_ELEMS + PG_STAGES * PG_BTILE_ELEMS) * sizeof(bf16_t);   // 96 + 64 = 160 KiB: all of a CU's LDS
  hipStream_t st = (hipStream_t)stream;
  // one process drives one GPU, but the CU count is still looked up per device
  static int n_cu_dev[64] = {0};
  int dev = 0;
  MAAVSS_CHECK_ARG(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, "vit_panel_gemm: cannot query the device");
  if (!n_cu_dev[dev]) {
    hipDeviceProp_t prop;
    MAAVSS_CHECK_ARG(hipGetDeviceProperties(&prop, dev) == hipSuccess, "vit_panel_gemm: cannot query the device");
    n_cu_dev[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int n_cu = n_cu_dev[dev];
  const int tail = g.panels % n_cu;        // one workgroup per CU (160 KiB of LDS): panels run in rounds of n_cu
  g.full = g.panels - tail;
  g.split = 1;
  if (tail > 0) {
    g.split = n_cu / tail;
    if (g.split > N / PG_BN) g.split = N / PG_BN;
    if (g.split < 1) g.split = 1;
  }
  const dim3 grid(g.full + tail * g.split), block(PG_THREADS);
  // hipFuncSetAttribute is per device and idempotent: set it at every launch (a few hundred ns) rather than cache a flag
#define PG_LAUNCH3(E, L, D)                                                                                          \
  {                                                                                                                  \
    hipFuncSetAttribute(reinterpret_cast<const void*>(vit_panel_gemm_kernel<E, L, D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL((vit_panel_gemm_kernel<E, L, D>), grid, block, smem, st, g);                                  \
  }
#define PG_LAUNCH(E, L) { if (dtype == MODE_F16) PG_LAUNCH3(E, L, MODE_F16) else PG_LAUNCH3(E, L, MODE_BF16) }
  if (X) {
    if (epilogue == 0) PG_LAUNCH(0, true) else if (epilogue == 1) PG_LAUNCH(1, true) else PG_LAUNCH(2, true)
  } else {
    if (epilogue == 0) PG_LAUNCH(0, false) else if (epilogue == 1) PG_LAUNCH(1, false) else PG_LAUNCH(2, false)
  }
#undef PG_LAUNCH3
#undef PG_LAUNCH
  MAAVSS_LAUNCH_CHECK("vit_panel_gemm_kernel");
  return MAAVSS_OK;
}
