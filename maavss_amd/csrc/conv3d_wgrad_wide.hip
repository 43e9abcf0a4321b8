// K8 (weight gradient), wide variant for the two large-M layers (Conv3d 16->32 and 32->64, reference
// avse_model_final.py:39,44).  The first wgrad kernel (conv3d.hip) gives every (kd,kh) its own workgroup, so each
// x / dy tile is pulled through L2 fifteen times (measured 14.3 GB fetched per launch for 16->32, MFMA busy 2 %).
// Here one 512-thread workgroup owns KDN*25 taps: all 75 taps for 16->32 (KDN = 3), one kd for 32->64 (KDN = 1),
// so a tile is staged once (resp. 3 times) and the 25..75 tap products run from the same LDS image:
//   dW[tap][ci][co] += sum over the tile's 256 positions of x[pos + tap][ci] * dy[pos][co]
// with the position as the MFMA K dimension, both operands read transposed from channels-last LDS tiles by
// ds_read_b64_tr_b16.  Output: the same partial layout as the first kernel ([chunk][tap][ci][co]) for the common
// deterministic reduce kernel.
#include "mma.h"

#ifndef WG_DEPTH
#define WG_DEPTH 3      // X16: (tap, channel block) pairs whose x fragments are read ahead of the MFMAs
#endif

// CIT > CI (64 -> 64 as two workgroups of 32 input channels each, blockIdx.y): x rows hold CIT channels, this workgroup stages
// and owns channels [CI blockIdx.y, +CI) -- every x byte is still staged once, dy (16-bit) twice; the one-(kd,kh)-per-workgroup
// kernel staged both fifteen times (0.83 ms per step for this layer, 2.4x its forward).
// 16 zero bytes: the source of LDS-DMA pieces that fall outside the image (X16)
__device__ const uint4 wgw_zero16 = {0u, 0u, 0u, 0u};

// X16 (round 4): x arrives as bf16 too (the producers bn_pool_act_fwd / conv3d_c1_bn_pool_act write that copy: the rounding this kernel's
// staging applied, done once) -- both images are then plain copies and go global -> LDS by DMA, 16 bytes per lane, into the OTHER of two image
// pairs while this tile's MFMAs run: no staging registers, no conversion, one barrier per tile.  Measured before: with the staging switched off
// the launches took 20-32 % less (scripts/wgrad_bench.py), register prefetch notwithstanding -- its ~400 vector instructions per tile and wave
// (addresses, bounds, conversion, LDS stores) compete with the MFMAs for issue at two waves per SIMD.
template <int MODE, int CI, int CO, int KDN, bool DY16 = false, int CIT = CI, bool X16 = false>
__global__ __launch_bounds__(512) void conv3d_wgrad_wide_kernel(const float* __restrict__ x, const void* __restrict__ dy_,
                                                               float* __restrict__ partials, int BT, int T, int H, int W,
                                                               int Ho, int Wo, int pad, int tiles_x, int tiles_y,
                                                               int tiles_per_chunk, int nchunk, int th) {
  using M = Mma<MODE>;
  using E = typename M::elem;
  constexpr int MT = CI / 16, NT = CO / 16, NTAP = KDN * 25, NPAIR = NTAP * MT, PW = (NPAIR + 7) / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // LDS images.  f32 mode: channels-last tiles xs [KDN][20 rows][20 cols][CI], ds [16][16][CO] (scalar reads).  16-bit modes (round 4): PLANAR by
  // 16-channel block -- xs [KDN][CI / 16][400 positions][16], ds [CO / 16][256 positions][16], planes XP / DP elements apart -- so that the
  // 32 lanes a ds_read_b64_tr_b16 serves together (8 positions x 4 lanes x 8 bytes) read 256 CONTIGUOUS bytes = every bank once.  In the
  // channels-last image a fragment touched 32 of every 64 (CI, CO = 32) or 128 (CO = 64) bytes and the lane groups G = 0 / 1 sat exactly 256 bytes
  // apart: 2-way conflicts on every read, 46-60 % of the kernel's LDS cycles (profiles/r3_e_kernel_pmc.json).  The planes are padded by 64 bytes so
  // that the staging writes of one position's blocks (plane stride = 0 mod 256 otherwise) spread over the banks too.
  constexpr bool PLANAR = MODE != MODE_F32;
  static_assert(!X16 || (MODE == MODE_BF16 && DY16), "X16: bf16 x and bf16 dy");
  constexpr int XP = X16 ? 400 * 16 : 400 * 16 + 32, DP = X16 ? 256 * 16 : 256 * 16 + 32;      // DMA writes lane-linearly: contiguous planes
  constexpr int BUF = X16 ? KDN * MT * XP + NT * DP : 0;                                       // elements per image pair (X16: two pairs)
  E* xs0 = reinterpret_cast<E*>(smem);
  E* ds0 = xs0 + (PLANAR ? KDN * MT * XP : KDN * 400 * CI);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int G = lane >> 4, l16 = lane & 15;
  const bool last_pair = wv + 8 * (PW - 1) < NPAIR;      // wave-uniform
  constexpr int KDG = 3 / KDN;               // kd groups (blockIdx.x % KDG)
  // XCD-aware mapping: the KDG blocks of one chunk take consecutive slots of one XCD (shared tiles hit its L2), and
  // each XCD owns a CONTIGUOUS eighth of the chunks: a chunk is about one frame of tiles, so the chunks that re-read
  // its frames (t-1, t+1: the kd planes) run next to it on the same L2 (PMC before: 1.85x the algorithmic bytes).
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int chunk = xcd * ((nchunk + 7) / 8) + slot / KDG, kdg = slot % KDG;
  if (chunk >= nchunk) return;
  const int kd0 = kdg * KDN;
  const int ci0 = CIT > CI ? (int)blockIdx.y * CI : 0;
  f32x4 acc[PW][NT];
#pragma unroll
  for (int p = 0; p < PW; ++p)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[p][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int tiles_total = BT * tiles_x * tiles_y;
  const int tile_beg = chunk * tiles_per_chunk, tile_end = min(tiles_total, tile_beg + tiles_per_chunk);
  // Staging is a REGISTER PREFETCH: all of a tile's global loads (NX + ND float4 per thread) are issued before the
  // previous tile's MFMAs and converted / written to LDS after them.  With one load in flight per thread (a plain
  // load -> convert -> store loop) the 13 dependent round trips per tile were 52 % of this kernel's time (scratch build
  // without staging: 2.2 vs 3.3 ms per step over the wgrad launches); one 512-thread workgroup per CU leaves 256
  // VGPRs per lane, so the 56 - 60 staging registers are free.
  // DY16: dy is stored in the operand format (bf16, by bn_pool_act_bwd): 8 elements per 16-byte vector, copied unconverted
  constexpr int DVE = DY16 ? 8 : 4;
  constexpr int XV = KDN * 400 * (CI / 4), DV = 256 * (CO / DVE);   // 16-byte vectors per tile
  constexpr int NX = (XV + 511) / 512, ND = (DV + 511) / 512;
  float4 xr[X16 ? 1 : NX], dr[X16 ? 1 : ND];
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    if constexpr (X16) return;
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, bt = tile / (tiles_x * tiles_y);
    const int t = bt % T, x0 = tx * 16, y0 = ty * th;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int i = tid + j * 512;
      const int c4 = (i % (CI / 4)) * 4, pos = (i / (CI / 4)) % 400, kdl = i / ((CI / 4) * 400);
      const int r = pos / 20, c = pos % 20;
      const int tt = t + kd0 + kdl - 1, iy = y0 + r - pad, ix = x0 + c - pad;
      xr[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < XV && tt >= 0 && tt < T && iy >= 0 && iy < H && ix >= 0 && ix < W)
        xr[j] = *reinterpret_cast<const float4*>(x + (((int64_t)(bt + kd0 + kdl - 1) * H + iy) * W + ix) * CIT + ci0 + c4);
    }
#pragma unroll
    for (int j = 0; j < ND; ++j) {
      const int i = tid + j * 512;
      const int pos = i / (CO / DVE), cv = (i % (CO / DVE)) * DVE;
      const int oy = y0 + pos / 16, ox = x0 + pos % 16;
      dr[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < DV && pos / 16 < th && oy < Ho && ox < Wo) {      // rows below the tile (th = 14): zeros
        const int64_t e = (int64_t)bt * Ho * Wo * CO + ((int64_t)oy * Wo + ox) * CO + cv;
        if constexpr (DY16) dr[j] = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned short*>(dy_) + e);
        else dr[j] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dy_) + e);
      }
    }
  };
  auto stash = [&]() __attribute__((always_inline)) {
    if constexpr (X16) return;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int i = tid + j * 512;
      if (i < XV) {
        E* d;
        if constexpr (PLANAR) {
          const int c4 = (i % (CI / 4)) * 4, pos = (i / (CI / 4)) % 400, kdl = i / ((CI / 4) * 400);
          d = xs0 + (kdl * MT + (c4 >> 4)) * XP + pos * 16 + (c4 & 15);
        } else {
          d = xs0 + (int64_t)i * 4;   // [kdl][pos][CI]: the float4 index is the element index / 4
        }
        d[0] = M::cvt(xr[j].x); d[1] = M::cvt(xr[j].y); d[2] = M::cvt(xr[j].z); d[3] = M::cvt(xr[j].w);
      }
    }
#pragma unroll
    for (int j = 0; j < ND; ++j) {
      const int i = tid + j * 512;
      if (i < DV) {
        E* d;
        if constexpr (PLANAR) {
          const int pos = i / (CO / DVE), cv = (i % (CO / DVE)) * DVE;
          d = ds0 + (cv >> 4) * DP + pos * 16 + (cv & 15);
        } else {
          d = ds0 + (int64_t)i * DVE;   // [pos][CO]
        }
        if constexpr (DY16) {
          *reinterpret_cast<float4*>(d) = dr[j];
        } else {
          d[0] = M::cvt(dr[j].x); d[1] = M::cvt(dr[j].y); d[2] = M::cvt(dr[j].z); d[3] = M::cvt(dr[j].w);
        }
      }
    }
  };
  // ---- X16: the DMA stream.  Piece i of the x image = 16 bytes = channels 8 (i & 1) .. + 7 of 16-channel block cb of position (i >> 1) % 400,
  // planes in (kdl, cb) order; of the dy image: the same with 256 positions per plane.  Thread-constant parts are computed once.
  constexpr int XPC = KDN * MT * 800, DPC = NT * 512, NXD = (XPC + 511) / 512, NDD = DPC / 512;
  int xo[X16 ? NXD : 1], xg[X16 ? NXD : 1], dyo[X16 ? NDD : 1], dg[X16 ? NDD : 1];      // element offset from the tile origin; packed (row, column, kdl)
  const unsigned short* x16 = reinterpret_cast<const unsigned short*>(x);
  const unsigned short* dy16 = reinterpret_cast<const unsigned short*>(dy_);
  if constexpr (X16) {
#pragma unroll
    for (int j = 0; j < NXD; ++j) {
      const int i = tid + 512 * j, plane = i / 800, rem = i % 800, pos = rem >> 1, r = pos / 20, c = pos % 20;
      const int kdl = plane / MT, cb = plane % MT;
      xo[j] = ((kdl * H + r) * W + c) * CIT + cb * 16 + (rem & 1) * 8;
      xg[j] = r | (c << 8) | (kdl << 16);
    }
#pragma unroll
    for (int j = 0; j < NDD; ++j) {
      const int i = tid + 512 * j, plane = i / 512, rem = i % 512, pos = rem >> 1;
      dyo[j] = ((pos >> 4) * Wo + (pos & 15)) * CO + plane * 16 + (rem & 1) * 8;
      dg[j] = (pos >> 4) | ((pos & 15) << 8);
    }
  }
  auto dma = [&](int tile, int buf) __attribute__((always_inline)) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, bt = tile / (tiles_x * tiles_y);
    const int t = bt % T, x0 = tx * 16, y0 = ty * th;
    const unsigned short* zeros = reinterpret_cast<const unsigned short*>(&wgw_zero16);
    const unsigned short* xb = x16 + (((int64_t)(bt + kd0 - 1) * H + (y0 - pad)) * W + (x0 - pad)) * CIT + ci0;
    E* xd = xs0 + buf * BUF;
#pragma unroll
    for (int j = 0; j < NXD; ++j) {
      const int i = tid + 512 * j;
      if (NXD * 512 == XPC || i < XPC) {
        const int r = xg[j] & 255, c = (xg[j] >> 8) & 255, kdl = xg[j] >> 16;
        const int tt = t + kd0 + kdl - 1, iy = y0 + r - pad, ix = x0 + c - pad;
        const unsigned short* src = (tt >= 0 && tt < T && iy >= 0 && iy < H && ix >= 0 && ix < W) ? xb + xo[j] : zeros;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(xd + (int64_t)i * 8), 16, 0, 0);
      }
    }
    const unsigned short* db = dy16 + (((int64_t)bt * Ho + y0) * Wo + x0) * CO;
    E* dd = ds0 + buf * BUF;
#pragma unroll
    for (int j = 0; j < NDD; ++j) {
      const int i = tid + 512 * j;
      const int oy = y0 + (dg[j] & 255), ox = x0 + (dg[j] >> 8);
      const unsigned short* src = ((dg[j] & 255) < th && oy < Ho && ox < Wo) ? db + dyo[j] : zeros;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dd + (int64_t)i * 8), 16, 0, 0);
    }
  };
  if constexpr (X16) {
    if (tile_beg < tile_end) dma(tile_beg, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  } else {
    if (tile_beg < tile_end) fetch(tile_beg);
  }
  for (int tile = tile_beg; tile < tile_end; ++tile) {
    const int cur = X16 ? (tile - tile_beg) & 1 : 0;
    const E* xs = xs0 + cur * BUF;
    const E* ds = ds0 + cur * BUF;
    if constexpr (X16) {
      if (tile + 1 < tile_end) dma(tile + 1, cur ^ 1);      // the other pair: last read by tile - 1, every wave is past the barrier that ended it
    } else {
      __syncthreads();   // the previous tile's fragment reads are done
      stash();
      __syncthreads();
      if (tile + 1 < tile_end) fetch(tile + 1);
    }
    // K step = 32 positions of the tile: rows 2 ks, 2 ks + 1 x 16 columns -- or, when at most 8 columns of the tile are inside the image (the last
    // tile column of a 56-wide plane), rows 4 ks .. 4 ks + 3 x 8 columns: half the steps (round 4).  th = 14: 7 / 4 steps (the dy rows below
    // the tile are staged as zeros; the x image always holds 20 rows).
    const int tx_cur = tile % tiles_x;
    const bool half = PLANAR && Wo - tx_cur * 16 <= 8;      // wave-uniform
    const int nks = half ? (th + 3) / 4 : th / 2, rstep = half ? 4 : 2, hrow = half ? 2 : 0, hcol = half ? 0 : 8;
#pragma unroll 1
    for (int ks = 0; ks < nks; ++ks) {
      typename M::frag fb[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if constexpr (MODE == MODE_F32) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int k = 8 * G + e;
            const float v = ds[((2 * ks + (k >> 4)) * 16 + (k & 15)) * CO + j * 16 + l16];
            if (e < 4) fb[j].lo[e] = v; else fb[j].hi[e - 4] = v;
          }
        } else {
          // MFMA k slot (G, hh, q = l16 >> 2) <-> tile position (row 2 ks + (G >> 1), column 4 (G & 1) + 8 hh + q): any assignment is valid as
          // long as both operands use it; this one gives the lanes 0-31 of a read (G = 0, 1) eight CONSECUTIVE positions
          bf16x4 h[2];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const int row = rstep * ks + (G >> 1) + hh * hrow, col = 4 * (G & 1) + hh * hcol + (l16 >> 2);
            const E* a = ds + j * DP + (row * 16 + col) * 16 + (l16 & 3) * 4;
            h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a));
          }
          fb[j] = concat4(h[0], h[1]);
        }
      }
      // x fragments of this wave's (tap, input-channel block) pairs.  Only the last pair slot of a wave can be empty (8 (PW - 1) <= NPAIR).
      static_assert(8 * (PW - 1) <= NPAIR, "pair slots");
      auto load_a = [&](int p) __attribute__((always_inline)) {
        const int q = wv + 8 * p;
        const int tap = q / MT, mi = q % MT;
        const int kdl = tap / 25, kh = (tap % 25) / 5, kw = tap % 5;
        const E* xb = PLANAR ? xs + (kdl * MT + mi) * XP : xs + kdl * 400 * CI + mi * 16;
        typename M::frag fa;
        if constexpr (MODE == MODE_F32) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int k = 8 * G + e;
            const float v = xb[((2 * ks + (k >> 4) + kh) * 20 + (k & 15) + kw) * CI + l16];
            if (e < 4) fa.lo[e] = v; else fa.hi[e - 4] = v;
          }
        } else {
          bf16x4 h[2];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const int row = rstep * ks + (G >> 1) + hh * hrow + kh, col = 4 * (G & 1) + hh * hcol + (l16 >> 2) + kw;
            const E* a = xb + (row * 20 + col) * 16 + (l16 & 3) * 4;
            h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a));
          }
          fa = concat4(h[0], h[1]);
        }
        return fa;
      };
      if constexpr (X16) {
        // the registers the DMA staging freed hold WG_DEPTH pairs of read-ahead: without it every pair is read -> wait -> NT MFMAs
        constexpr int DEPTH = WG_DEPTH;
        typename M::frag fa[PW];
#pragma unroll
        for (int p = 0; p < DEPTH && p < PW; ++p)
          if (p < PW - 1 || last_pair) fa[p] = load_a(p);
#pragma unroll
        for (int p = 0; p < PW; ++p) {
          if (p + DEPTH < PW && (p + DEPTH < PW - 1 || last_pair)) fa[p + DEPTH] = load_a(p + DEPTH);
          __builtin_amdgcn_sched_barrier(0);
          if (p < PW - 1 || last_pair) {
#pragma unroll
            for (int j = 0; j < NT; ++j) M::mma(acc[p][j], fa[p], fb[j]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int p = 0; p < PW; ++p) {
          if (p < PW - 1 || last_pair) {  // wave-uniform
            const typename M::frag fa = load_a(p);
#pragma unroll
            for (int j = 0; j < NT; ++j) M::mma(acc[p][j], fa, fb[j]);
          }
        }
      }
    }
    if constexpr (X16) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the next tile have landed
      __syncthreads();
    }
  }
  // partials[chunk][tap = (kd*5+kh)*5+kw][ci][co]
#pragma unroll
  for (int p = 0; p < PW; ++p) {
    const int q = wv + 8 * p;
    if (q < NPAIR) {
      const int tap = q / MT, mi = q % MT;
      float* out = partials + (((int64_t)chunk * 75 + kd0 * 25 + tap) * CIT + ci0 + mi * 16 + G * 4) * CO;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(int64_t)r * CO + j * 16 + l16] = acc[p][j][r];
    }
  }
}

int maavss_conv_tile_h(int Ho);      // conv3d.hip

template <int MODE, int CI, int CO, int KDN, bool DY16 = false, int CIT = CI, bool X16 = false>
static void launch_wide(const float* x, const void* dy, float* ws, int BT, int T, int H, int W, int Ho, int Wo, int pad,
                        int nchunk, hipStream_t st) {
  using E = typename Mma<MODE>::elem;
  const size_t smem = MODE == MODE_F32 ? (KDN * 400 * CI + 256 * CO) * sizeof(E)
                      : X16        ? 2 * (size_t)(KDN * (CI / 16) * 400 * 16 + (CO / 16) * 256 * 16) * sizeof(E)
                                   : (size_t)(KDN * (CI / 16) * (400 * 16 + 32) + (CO / 16) * (256 * 16 + 32)) * sizeof(E);
  auto kern = conv3d_wgrad_wide_kernel<MODE, CI, CO, KDN, DY16, CIT, X16>;
  if (smem > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  // tile height 14 where that covers the plane with as many tiles as 16 would (conv3d.hip, conv_tile_h): 56 -> 4 x 14, 28 -> 2 x 14
  const int th = MODE != MODE_F32 ? maavss_conv_tile_h(Ho) : 16;
  const int tiles_x = cdiv(Wo, 16), tiles_y = cdiv(Ho, th);
  const int tiles_total = BT * tiles_x * tiles_y;
  const int tpc = cdiv(tiles_total, nchunk);
  constexpr int KDG = 3 / KDN;
  hipLaunchKernelGGL(kern, dim3(KDG * cdiv(nchunk, 8) * 8, CIT / CI), dim3(512), smem, st, x, dy, ws, BT, T, H, W, Ho, Wo, pad, tiles_x,
                     tiles_y, tpc, nchunk, th);
}

// returns 1 if this (c_in, c_out) pair is handled by the wide kernel (and launches it), 0 otherwise
int maavss_conv3d_wgrad_wide_try(const float* x, const void* dy, float* ws, int nchunk, int B, int T, int H, int W, int Ho,
                                 int Wo, int c_in, int c_out, int pad, int mode, int dy16, int x16, hipStream_t st) {
#define WIDE(CI, CO, KDN)                                                                                      \
  if (c_in == CI && c_out == CO) {                                                                             \
    if (mode == MODE_F32) launch_wide<MODE_F32, CI, CO, KDN>(x, dy, ws, B * T, T, H, W, Ho, Wo, pad, nchunk, st);      \
    else if (mode == MODE_F16) launch_wide<MODE_F16, CI, CO, KDN>(x, dy, ws, B * T, T, H, W, Ho, Wo, pad, nchunk, st); \
    else if (dy16 && x16) launch_wide<MODE_BF16, CI, CO, KDN, true, CI, true>(x, dy, ws, B * T, T, H, W, Ho, Wo, pad, nchunk, st); \
    else if (dy16) launch_wide<MODE_BF16, CI, CO, KDN, true>(x, dy, ws, B * T, T, H, W, Ho, Wo, pad, nchunk, st);      \
    else launch_wide<MODE_BF16, CI, CO, KDN>(x, dy, ws, B * T, T, H, W, Ho, Wo, pad, nchunk, st);                      \
    return 1;                                                                                                  \
  }
  WIDE(16, 32, 3)
  WIDE(32, 64, 1)
#undef WIDE
  if (c_in == 64 && c_out == 64 && mode == MODE_BF16) {       // two 32-channel halves of x per tile (16-bit path only)
    if (dy16 && x16) launch_wide<MODE_BF16, 32, 64, 1, true, 64, true>(x, dy, ws, B * T, T, H, W, Ho, Wo, pad, nchunk, st);
    else if (dy16) launch_wide<MODE_BF16, 32, 64, 1, true, 64>(x, dy, ws, B * T, T, H, W, Ho, Wo, pad, nchunk, st);
    else launch_wide<MODE_BF16, 32, 64, 1, false, 64>(x, dy, ws, B * T, T, H, W, Ho, Wo, pad, nchunk, st);
    return 1;
  }
  return 0;
}
