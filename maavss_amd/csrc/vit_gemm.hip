// K1/K3: the dense layers of the DINO ViT-S/8 (patch projection, qkv, proj, MLP fc1/fc2) as ONE bf16 MFMA GEMM
// with fused epilogues.  Replaces the Linear/Conv2d calls inside dino's VisionTransformer that the reference
// reaches through video_attention.py:52 (get_last_selfattention); the module is external to the reference
// (empty submodule), see oracle/vit_ref_cpu.py for the restated architecture.
//
//   C[M,N] = epilogue( A[M,K] (bf16, row-major) . W[N,K]^T (bf16, torch Linear layout) )
//
// 256x128 output tile per 512-thread workgroup (4x2 waves, 64x64 per wave = 4x4 MFMA 16x16x32 tiles), BK = 64.
// These GEMMs have SHORT K (384 / 1536 / 192): what limits them is the L2 -> LDS feed, not MFMA issue
// (measured: a 128x128 / BK=32 version spent 88 % of its time with the MFMAs removed, because every 128-B line
// was requested twice as two 64-B halves).  Hence:
//   * BK = 64 bf16 = one full 128-B line per row and staging instruction; 256x128 tile = 85 FLOP per staged byte;
//   * operand tiles go global -> LDS directly (global_load_lds, 16 B per lane) into a 3-slot ring (144 KiB, one
//     workgroup per CU); two K-steps stay in flight across a raw s_barrier behind a counted vmcnt;
//   * the DMA instructions are issued by FOUR DEDICATED LOADER WAVES (waves 8-11: 12 one-KiB pieces each per K-step),
//     not by the eight MFMA waves: one global_load_lds costs its wave 60-300 issue cycles ("What the
//     K = 384 GEMM taught"), which with the stream spread over the MFMA waves came out of the MFMA issue slots;
//     the loaders execute exactly the barrier sequence of the MFMA waves;
//   * the workgroups are PERSISTENT (one per CU) and the DMA stream runs across tile boundaries: while a tile's
//     epilogue runs (staged through the ring slot that just became free), the first two K-steps of the next
//     tile are already in flight, so the per-tile prologue latency -- the dominant cost at K = 384 -- is hidden;
//   * the LDS fragment reads of the next 32-deep sub-step are issued BEFORE the 16 MFMAs of the current one (two
//     named fragment sets), so LDS latency hides under the matrix pipe;
//   * the LDS image is lane-linear, so the bank swizzle (16-B chunk ^= row & 7) is applied to the per-lane SOURCE
//     address and again on the ds_read_b128 side;
//   * blocks are remapped so that the N-tiles of one M-panel run on the same XCD (A panel from that L2).
// Epilogues: +bias (and log2e/8 on the q third) -> bf16 | +bias, GELU -> bf16 | +bias +residual -> f32 (in place
// on the residual stream) | + periodic row table (conv bias / cls token + position embedding) -> f32.
// All outputs are staged through LDS and written as full rows (16 B per lane).
#include "mma.h"

#define EPI_BF16_BIAS 0
#define EPI_BF16_BIAS_GELU 1
#define EPI_F32_BIAS_RESID 2
#define EPI_F32_ROWTABLE 3

#define VG_BM 256
#define VG_BN 128
#define VG_BK 64
#define VG_STAGES 3
#define VG_MMA_WAVES 8
#define VG_LOADERS 4
#define VG_THREADS 768
#define VG_MMA_THREADS 512
#define VG_STAGE_ELEMS ((VG_BM + VG_BN) * VG_BK)  // A tile + B tile, bf16 elements (48 KiB)
#define VG_LOADS 12                                // global_load_lds per LOADER wave per stage (8 A + 4 B)

struct VGemmArgs {
  const bf16_t* A;
  const bf16_t* W;
  const float* bias;      // [N] (EPI 0..2)
  const float* table;     // [period][N] (EPI 3)
  void* C;                // bf16 [M][ldc] or f32 [M][ldc]
  int M, N, K, lda, ldc;
  int period;             // EPI 3: row r uses table[r % period]
  int qscale_cols;        // EPI 0: columns < qscale_cols are multiplied by qscale after the bias
  float qscale;
  int tiles_n, tiles_m;
  float* row_stats;       // f32 epilogues, optional: [M][tiles_n][2] = (mean, sum of squared deviations) of each row's 128 tile columns
};

__device__ __forceinline__ void glds16(const void* g, void* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// exact-erf GELU with erf from Abramowitz-Stegun 7.1.26 (|abs err| < 1.5e-7, far below the bf16 output rounding)
__device__ __forceinline__ float gelu_erf(float v) {
  const float x = fabsf(v) * 0.70710678118654752f;
  const float t = __frcp_rn(1.f + 0.3275911f * x);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.f - poly * __expf(-x * x);
  const float erf_v = v < 0.f ? -erf_abs : erf_abs;
  return 0.5f * v * (1.f + erf_v);
}

template <int EPI, int MODE>
__global__ __launch_bounds__(VG_THREADS) void vit_gemm_kernel(VGemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* lds = reinterpret_cast<bf16_t*>(smem);  // ring: [3 slots][A 256x64 | B 128x64] bf16 = 144 KiB
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int l16 = lane & 15, gq = lane >> 4;
  const int nk = g.K / VG_BK;

  // ---- persistent tile schedule.  Blocks are dealt round-robin over the 8 XCDs; XCD x owns a contiguous range
  // of tiles (tn fastest), its blocks take them round-robin, so co-running blocks of one XCD share A panels in L2.
  const int ntiles = g.tiles_m * g.tiles_n;
  const int xcd = blockIdx.x & 7, lane_in_xcd = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const int q8 = ntiles / 8, r8 = ntiles % 8;
  const int lo = xcd * q8 + min(xcd, r8), hi = lo + q8 + (xcd < r8 ? 1 : 0);
  const int my_first = lo + lane_in_xcd;
  const int my_count = my_first < hi ? (hi - my_first + per_xcd - 1) / per_xcd : 0;
  if (my_count == 0) return;
  const int total_steps = my_count * nk;  // virtual K-step stream across this block's tiles

  // ---- staging stream (loader waves only; runs two K-steps ahead of the MFMAs, across tile boundaries).
  // Loader lw stages A rows lw*64 .. +63 (8 pieces of 8 rows) and W rows lw*32 .. +31 (4 pieces).
  const int srow = lane >> 3, sslot = lane & 7, lw = wv - VG_MMA_WAVES;
  const bf16_t* asrc[8];
  const bf16_t* bsrc[4];
  int s_tile = -1;
  auto set_stage_tile = [&](int ti) {
    const int id = my_first + ti * per_xcd;
    const int m0 = (id / g.tiles_n) * VG_BM, n0 = (id % g.tiles_n) * VG_BN;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = lw * 64 + i * 8 + srow;
      int ar = m0 + r;
      ar = ar < g.M ? ar : g.M - 1;
      asrc[i] = g.A + (int64_t)ar * g.lda + ((sslot ^ (r & 7)) * 8);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = lw * 32 + i * 8 + srow;
      bsrc[i] = g.W + (int64_t)(n0 + r) * g.K + ((sslot ^ (r & 7)) * 8);
    }
    s_tile = ti;
  };
  auto stage = [&](int v) {  // virtual step v -> ring slot v % 3
    const int ti = v / nk, kt = v - ti * nk;
    if (ti != s_tile) set_stage_tile(ti);
    bf16_t* la = lds + (v % VG_STAGES) * VG_STAGE_ELEMS;
    bf16_t* lb = la + VG_BM * VG_BK;
#pragma unroll
    for (int i = 0; i < 8; ++i) glds16(asrc[i] + kt * VG_BK, la + (lw * 64 + i * 8) * VG_BK + lane * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(bsrc[i] + kt * VG_BK, lb + (lw * 32 + i * 8) * VG_BK + lane * 8);
  };
  // barriers of one tile epilogue (the loaders execute them too)
  constexpr int EPI_BARRIERS = (EPI == EPI_BF16_BIAS || EPI == EPI_BF16_BIAS_GELU) ? 3 : 7;

  if (wv >= VG_MMA_WAVES) {
    // ================= loader waves: the MFMA waves' barrier sequence with the DMA issue in between =================
    stage(0);
    if (total_steps > 1) stage(1);
    if (total_steps > 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // step 0 landed (this wave's part)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int v = 0;
    for (int ti = 0; ti < my_count; ++ti) {
      for (int kt = 0; kt < nk; ++kt, ++v) {
        // slot of step v-1 is free (all fragment reads of step v-1 retired before the last barrier) -> refill it
        const bool issued = v + 2 < total_steps;
        if (issued) stage(v + 2);
        if (kt + 1 < nk) {
          // make step v+1 visible: counted vmcnt (the step issued above may stay in flight) + barrier
          if (issued) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
      }
      __builtin_amdgcn_s_barrier();   // tile done
#pragma unroll
      for (int b = 0; b < EPI_BARRIERS; ++b) __builtin_amdgcn_s_barrier();
      if (ti + 1 < my_count) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile's first two steps have landed
        __builtin_amdgcn_s_barrier();
      }
    }
    return;
  }

  int rowa[4], rowb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    rowa[i] = wm * 64 + i * 16 + l16;
    rowb[i] = wn * 64 + i * 16 + l16;
  }
  auto load_frags = [&](int slot, int s, bf16x8 (&fa)[4], bf16x8 (&fb)[4]) {
    const bf16_t* la = lds + slot * VG_STAGE_ELEMS;
    const bf16_t* lb = la + VG_BM * VG_BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(la + rowa[i] * VG_BK + (((s * 4 + gq) ^ (rowa[i] & 7)) * 8));
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(lb + rowb[j] * VG_BK + (((s * 4 + gq) ^ (rowb[j] & 7)) * 8));
  };
  f32x4 acc[4][4];
  auto mfmas = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) Mma<MODE>::mma(acc[i][j], fa[i], fb[j]);
  };
#define VG_USE_FRAGS(fa, fb)                 \
  __builtin_amdgcn_sched_barrier(0);         \
  asm volatile("" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]))

  // ================= MFMA waves =================
  bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];
  __builtin_amdgcn_s_barrier();   // step 0 landed (loaders waited for it)
  load_frags(0, 0, fa0, fb0);
  VG_USE_FRAGS(fa0, fb0);

  int v = 0;
  for (int ti = 0; ti < my_count; ++ti) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; ++kt, ++v) {
      const int slot = v % VG_STAGES;
      load_frags(slot, 1, fa1, fb1);   // sub-step 1 reads go out first, sub-step 0 MFMAs run underneath
      mfmas(fa0, fb0);
      VG_USE_FRAGS(fa1, fb1);
      if (kt + 1 < nk) {
        __builtin_amdgcn_s_barrier();   // step v+1 visible (the loaders waited for it), slot v-1 handed back
        load_frags((v + 1) % VG_STAGES, 0, fa0, fb0);
      }
      mfmas(fa1, fb1);
      if (kt + 1 < nk) { VG_USE_FRAGS(fa0, fb0); }
    }
    // ---- tile done.  Steps v, v+1 (next tile) are already in flight in the other two slots; the slot of the last
    // step (v-1) is free and serves as the epilogue staging buffer, so the next tile's prologue latency hides here.
    __builtin_amdgcn_s_barrier();   // every wave has finished its fragment reads of the last step
    {
      const int id = my_first + ti * per_xcd;
      const int m0 = (id / g.tiles_n) * VG_BM, n0 = (id % g.tiles_n) * VG_BN;
      char* ebuf = smem + ((v - 1) % VG_STAGES) * (VG_STAGE_ELEMS * 2);
      if constexpr (EPI == EPI_BF16_BIAS || EPI == EPI_BF16_BIAS_GELU) {
        constexpr int LDC = 136;  // [128][136] bf16 = 34 KiB per pass
        bf16_t* cs = reinterpret_cast<bf16_t*>(ebuf);
        bf16_t* C = reinterpret_cast<bf16_t*>(g.C);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          if (pass) __syncthreads();
          if ((wm >> 1) == pass) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int nl = wn * 64 + j * 16 + l16;
              const int n = n0 + nl;
              const float bv = g.bias[n];
              const float sc = (EPI == EPI_BF16_BIAS && n < g.qscale_cols) ? g.qscale : 1.f;
#pragma unroll
              for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  float val = acc[i][j][r] + bv;
                  if constexpr (EPI == EPI_BF16_BIAS_GELU) val = gelu_erf(val);
                  else val *= sc;
                  cs[((wm & 1) * 64 + i * 16 + gq * 4 + r) * LDC + nl] = cvt16<MODE>(val);
                }
            }
          }
          __syncthreads();
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int q = it * VG_MMA_THREADS + tid, row = q >> 4, c16 = q & 15;
            const int m = m0 + pass * 128 + row;
            if (m < g.M)
              *reinterpret_cast<uint4*>(C + (int64_t)m * g.ldc + n0 + c16 * 8) = *reinterpret_cast<const uint4*>(cs + row * LDC + c16 * 8);
          }
        }
      } else {
        constexpr int LDF = 132;  // [64][132] f32 = 33 KiB per pass
        float* cf = reinterpret_cast<float*>(ebuf);
        float* C = reinterpret_cast<float*>(g.C);
        // The residual (or row-table) values are requested AHEAD of their use -- passes 0-1 up front, pass 2 / 3 as pass
        // 0 / 1 release their registers (8 + 4 float4 per thread in flight, in the registers the operand fragments just
        // vacated; the budget is 168 VGPRs at 12 waves per CU) -- so their HBM latency is paid about once per tile
        // instead of once per pass (one workgroup per CU: nothing else would cover it; it was 240 of 710 us on fc2;
        // requesting them under the tile's last MFMAs as well spills).  The barriers here are raw s_barriers behind an
        // LDS-only wait: __syncthreads() would also drain vmcnt, i.e. wait for these loads and the previous stores.
        const int c4 = (tid & 31) * 4, rbase = tid >> 5;
        // addresses: a buffer descriptor on the tile (wave-uniform, SGPRs) + one 32-bit lane offset (the 16-row step is
        // added at each use: the range check covers the VGPR offset, not an SGPR one), so the sixteen requests hold no
        // address registers; the descriptor ends after the last real row: rows past M read 0 and are not written.
        typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
        unsigned voff = (unsigned)(rbase * g.ldc + c4) * 4u;
        asm volatile("" : "+v"(voff));   // per tile: hipcc otherwise hoists the sixteen offsets out of the tile loop and spills them
        const unsigned cstep = 16u * (unsigned)g.ldc * 4u;
        const int64_t rem = ((int64_t)(g.M - m0 - 1) * g.ldc + VG_BN) * 4;
        const __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(
            C + (int64_t)m0 * g.ldc + n0, 0, (int)(rem < 0xFFFFFFF0LL ? rem : 0xFFFFFFF0LL), 0x00020000);
        float4 addv[16];
        auto request = [&](int e) __attribute__((always_inline)) {   // e = pass * 4 + it: row pass*64 + it*16 + (tid >> 5)
          if constexpr (EPI == EPI_F32_BIAS_RESID) {
            addv[e] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(crsrc, voff + e * cstep, 0, 0));
          } else {
            const int m = m0 + e * 16 + rbase;
            addv[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < g.M) addv[e] = *reinterpret_cast<const float4*>(g.table + (int64_t)(m % g.period) * g.N + n0 + c4);
          }
        };
#pragma unroll
        for (int e = 0; e < 8; ++e) request(e);
        float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (EPI == EPI_F32_BIAS_RESID) bb = *reinterpret_cast<const float4*>(g.bias + n0 + c4);
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          if (pass) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
          }
          if (wm == pass) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) cf[(i * 16 + gq * 4 + r) * LDF + wn * 64 + j * 16 + l16] = acc[i][j][r];
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" : "+v"(voff) : : "memory");
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int row = it * 16 + rbase;
            const float4 a = *reinterpret_cast<const float4*>(cf + row * LDF + c4);
            const float4 res = addv[pass * 4 + it];
            float4 add;
            if constexpr (EPI == EPI_F32_BIAS_RESID) add = make_float4(res.x + bb.x, res.y + bb.y, res.z + bb.z, res.w + bb.w);
            else add = res;
            const float4 o4 = make_float4(a.x + add.x, a.y + add.y, a.z + add.z, a.w + add.w);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o4), crsrc, voff + (pass * 4 + it) * cstep, 0, 0);
            if (g.row_stats != nullptr) {
              // LayerNorm statistics of the stored row over this tile's 128 columns (32 consecutive lanes = one row): the
              // consumer (vit_ws_gemm with LayerNorm on the way in) merges the N / 128 partials of a row
              const float mt = half_wave_sum((o4.x + o4.y) + (o4.z + o4.w)) * (1.f / VG_BN);
              const float dx = o4.x - mt, dy = o4.y - mt, dz = o4.z - mt, dw = o4.w - mt;
              const float m2 = half_wave_sum((dx * dx + dy * dy) + (dz * dz + dw * dw));
              const int m = m0 + pass * 64 + row;
              if ((tid & 31) == 0 && m < g.M)
                *reinterpret_cast<float2*>(g.row_stats + ((int64_t)m * g.tiles_n + n0 / VG_BN) * 2) = make_float2(mt, m2);
            }
          }
          if (pass < 2) {
#pragma unroll
            for (int e = 8 + 4 * pass; e < 12 + 4 * pass; ++e) request(e);
          }
        }
      }
    }
    if (ti + 1 < my_count) {
      // The staging slot is free again (every wave's LDS reads are done) and the loaders have waited for the next tile's
      // first two steps; this tile's global stores keep draining under the next tile's MFMAs (no vmcnt wait here).
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      load_frags(v % VG_STAGES, 0, fa0, fb0);
      VG_USE_FRAGS(fa0, fb0);
    }
  }
}

extern "C" int maavss_vit_gemm_stats(const void* A, int lda, const void* W, const float* bias, const float* table, int period,
                                     void* C, int ldc, int64_t M, int N, int K, int epilogue, int qscale_cols, float qscale,
                                     float* row_stats, int dtype, void* stream) {
  MAAVSS_CHECK_ARG(A && W && C && M > 0, "vit_gemm: bad arguments");
  MAAVSS_CHECK_ARG(N % VG_BN == 0 && K % VG_BK == 0 && K >= VG_BK, "vit_gemm: N must be a multiple of 128 and K of 64 (N=%d K=%d)", N, K);
  MAAVSS_CHECK_ARG(lda % 8 == 0 && ldc % 8 == 0, "vit_gemm: leading dimensions must be multiples of 8");
  MAAVSS_CHECK_ARG(epilogue >= 0 && epilogue <= 3, "vit_gemm: unknown epilogue");
  MAAVSS_CHECK_ARG(epilogue == EPI_F32_ROWTABLE ? (table && period > 0) : (bias != nullptr), "vit_gemm: missing bias/table");
  MAAVSS_CHECK_ARG(M < (1LL << 31), "vit_gemm: M too large");
  MAAVSS_CHECK_ARG(dtype == MODE_BF16 || dtype == MODE_F16, "vit_gemm: dtype must be 0 (bf16) or 2 (f16)");
  MAAVSS_CHECK_ARG(!row_stats || epilogue >= EPI_F32_BIAS_RESID, "vit_gemm: row statistics come with the f32 epilogues (2, 3) only");
  VGemmArgs g;
  g.row_stats = row_stats;
  g.A = (const bf16_t*)A; g.W = (const bf16_t*)W; g.bias = bias; g.table = table; g.C = C;
  g.M = (int)M; g.N = N; g.K = K; g.lda = lda; g.ldc = ldc; g.period = period;
  g.qscale_cols = qscale_cols; g.qscale = qscale;
  g.tiles_n = N / VG_BN; g.tiles_m = cdiv(M, VG_BM);
  // persistent: one workgroup per CU (144 KiB LDS each), a multiple of 8 so every XCD gets the same number
  int nblocks = g.tiles_n * g.tiles_m;
  nblocks = nblocks >= 256 ? 256 : cdiv(nblocks, 8) * 8;
  const dim3 grid(nblocks), block(VG_THREADS);
  const size_t smem = VG_STAGES * VG_STAGE_ELEMS * sizeof(bf16_t);   // 144 KiB
  hipStream_t st = (hipStream_t)stream;
  // the attribute is per device and idempotent: set at every launch instead of caching a process-wide flag
#define VG_LAUNCH2(E, D)                                                                                                       \
  {                                                                                                                            \
    hipFuncSetAttribute(reinterpret_cast<const void*>(vit_gemm_kernel<E, D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL((vit_gemm_kernel<E, D>), grid, block, smem, st, g);                                                     \
  }
#define VG_LAUNCH(E) { if (dtype == MODE_F16) VG_LAUNCH2(E, MODE_F16) else VG_LAUNCH2(E, MODE_BF16) }
  switch (epilogue) {
    case EPI_BF16_BIAS: VG_LAUNCH(EPI_BF16_BIAS) break;
    case EPI_BF16_BIAS_GELU: VG_LAUNCH(EPI_BF16_BIAS_GELU) break;
    case EPI_F32_BIAS_RESID: VG_LAUNCH(EPI_F32_BIAS_RESID) break;
    default: VG_LAUNCH(EPI_F32_ROWTABLE) break;
  }
#undef VG_LAUNCH
#undef VG_LAUNCH2
  MAAVSS_LAUNCH_CHECK("vit_gemm_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_vit_gemm(const void* A, int lda, const void* W, const float* bias, const float* table, int period,
                               void* C, int ldc, int64_t M, int N, int K, int epilogue, int qscale_cols, float qscale,
                               int dtype, void* stream) {
  return maavss_vit_gemm_stats(A, lda, W, bias, table, period, C, ldc, M, N, K, epilogue, qscale_cols, qscale, nullptr, dtype, stream);
}
