// K7/K8: the five Conv3d(k=(3,5,5), stride 1, pad (1,p,p), bias=False) layers of the visual encoder
// (reference avse_model_final.py:34,39,44,49,54) -- forward, input gradient and weight gradient.
//
// Activations are channels-last [B,T,H,W,C] f32 in HBM (C=1 for the network input, so the reference's
// NCDHW input is used as is).  Three kernel families:
//   conv3d_c1_*      C_in = 1 (K = 75): direct VALU convolution from an LDS halo tile, weights as
//                    wave-uniform scalar operands.
//   conv3d_igemm     C_in in {16,32,64}: implicit GEMM on MFMA.  One workgroup = a 16x16 tile of output
//                    positions of one (b,t) plane x all C_out.  For each kd the 20x20xC_in input halo is
//                    staged ONCE into LDS (XOR-swizzled 16-byte chunks) and re-read for the 25 (kh,kw)
//                    taps; weights stream through a double-buffered LDS tile of 64 k per step.  The same
//                    kernel computes the input gradient (flipped/transposed weights, pad 4-p).
//   conv3d_wgrad     dW = sum over positions of x^T . dy; position is the MFMA K dimension, read from
//                    channels-last LDS tiles with ds_read_b64_tr_b16 (hardware transpose).  A workgroup
//                    owns one (kd,kh) and the 5 kw taps, walks a chunk of position tiles and writes a
//                    partial; a second kernel sums the chunks (deterministic, no atomics).
#include <type_traits>
#include <utility>
#include <cstdlib>
#include "mma.h"

// --------------------------------------------------------------------------------------------
// 16-byte-chunk XOR swizzle for an LDS image with rows of RB bytes (RB = 32..256, power of two),
// so that 16 consecutive rows read at the same chunk hit 16 different 16-byte bank slots.
template <int RB>
__device__ __forceinline__ int swz(int row, int chunk) {
  constexpr int PPR = RB >= 256 ? 1 : 256 / RB;
  constexpr int NCH = RB / 16;
  return chunk ^ ((row / PPR) & (NCH - 1));
}

// Halo image of the implicit GEMM, 16-bit modes: chunk swizzle chosen for how ds_read_b128 is serviced -- four groups of 16 lanes,
// {0-3, 12-15, 20-27} etc. (MI355X_MICROARCH.md, LDS): a group holds the 16 positions of a fragment row, EIGHT of them with k
// sub-block g and eight with g + 1 (chunks c and c + 1 of a position).  The row-XOR above (chunk ^= column / PPR) makes every
// such read 2-way conflicted (PMC: 19-37 % of the LDS cycles of the igemm kernels were conflict cycles at 43-65 % LDS busy);
// a search over the linear maps of the column bits gives conflict-free ones: none for 2 chunks per position, bit 2 of the
// column into chunk bit 1 for 4 chunks, column bits 1-2 into chunk bits 1-2 for 8 chunks.  (XOR: the source-side swizzle of the
// LDS-DMA path is the same function.)
template <int RB, int ES>
__device__ __forceinline__ int swz_halo(int col, int chunk) {
  if constexpr (ES != 2) return swz<RB>(col, chunk);
  else if constexpr (RB == 32) return chunk;
  else if constexpr (RB == 64) return chunk ^ (((col >> 2) & 1) << 1);
  else if constexpr (RB == 128) return chunk ^ (col & 6);
  else return swz<RB>(col, chunk);
}

// --------------------------------------------------------------------------------------------
// weight re-layout: reference [CO][CI][3][5][5] f32  ->  wt[kd][n][KP] (k = (kh*5+kw)*CIN + ci), elem type.
//   mode 0 (forward): n = co, CIN = CI, value W[co][ci][kd][kh][kw]
//   mode 1 (dgrad)  : n = ci, CIN = CO, value W[co][ci][2-kd][4-kh][4-kw]
template <int PRECISE>
__global__ void conv3d_prep_w_kernel(const float* __restrict__ w, typename Mma<PRECISE>::elem* __restrict__ wt, int CO, int CI,
                                     int KP, int mode) {
  const int nN = mode ? CI : CO, cin = mode ? CO : CI;
  const int64_t total = 3LL * nN * KP;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int kk = (int)(i % KP);
    const int n = (int)((i / KP) % nN);
    const int kd = (int)(i / ((int64_t)KP * nN));
    float v = 0.f;
    if (kk < 25 * cin) {
      const int tap = kk / cin, c = kk % cin, kh = tap / 5, kw = tap % 5;
      if (!mode) v = w[(((int64_t)n * CI + c) * 3 + kd) * 25 + kh * 5 + kw];
      else v = w[(((int64_t)c * CI + n) * 3 + (2 - kd)) * 25 + (4 - kh) * 5 + (4 - kw)];
    }
    wt[i] = Mma<PRECISE>::cvt(v);
  }
}

// --------------------------------------------------------------------------------------------
// XCD-aware tile order for the 16x16-output-tile kernels.  Workgroups are dealt round-robin to the 8 XCDs, each with
// its own L2: with a plain (tx, ty, bt) grid the tiles that share input -- x / y neighbours (20x20 halo for a 16x16
// tile) and the same tile of frames t-1, t, t+1 (three kd planes) -- land on eight different L2s and every re-read goes
// to HBM (PMC before: 3.1 GB fetched for a 0.79 GB input by the 32->16 dgrad, 3.3-3.9x on the other layers).  Here
// XCD k walks the k-th contiguous eighth of the tile list, tx fastest, then ty, then bt, so those re-reads meet in L2.
struct TileId { int tx, ty, bt; int64_t lin; bool valid; };
__device__ __forceinline__ TileId xcd_tile(int nx, int ny, int64_t total64) {
  // 32-bit unsigned arithmetic (the host checks total < 2^31): every wave of a workgroup runs this on the CU's one scalar
  // unit, and the 64-bit divisions of the first version were ~300 scalar instructions per wave
  const unsigned total = (unsigned)total64, per = (total + 7) / 8;
  const unsigned lin = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  TileId t;
  t.valid = (blockIdx.x >> 3) < per && lin < total;
  t.lin = lin;
  const unsigned row = lin / (unsigned)nx;
  t.tx = (int)(lin - row * (unsigned)nx);
  t.bt = (int)(row / (unsigned)ny);
  t.ty = (int)(row - (unsigned)t.bt * (unsigned)ny);
  return t;
}
static inline int xcd_grid(int64_t total) { return (int)(((total + 7) / 8) * 8); }

// Tile HEIGHT of the 16-wide output tiles (round 4): 14 rows when that covers the plane with as many tiles as 16 would (56 -> 4 x 14, 28 -> 2 x 14:
// the layers at 56^2 and 28^2 spent 12.5 % of their MFMAs on rows below the image), else 16.  A tile's rows are dealt to the four waves as
// 4 + 4 + 3 + 3 (first row 0, 4, 8, 11); the number of tiles -- and of BatchNorm partial rows -- is the same for both heights by construction.
int maavss_conv_tile_h(int Ho) {      // (also conv3d_wgrad_wide.hip)
  static const bool only16 = getenv("MAAVSS_TILE_H16") != nullptr;      // A/B switch
  return !only16 && cdiv(Ho, 14) == cdiv(Ho, 16) ? 14 : 16;
}
static inline int conv_tile_h(int Ho) { return maavss_conv_tile_h(Ho); }
__device__ __forceinline__ int tile_row0(int wv, int th) { return th == 14 ? 4 * wv - (wv > 2 ? wv - 2 : 0) : 4 * wv; }
__device__ __forceinline__ int tile_nrows(int wv, int th) { return th == 14 && wv >= 2 ? 3 : 4; }

// IN16: x is already stored in the MFMA operand format (IEEE half for the forward pass, bf16 for the input-gradient pass:
// the producers bn_pool_act_fwd / bn_pool_act_bwd round once instead of every consumer) -- the halo is then a plain copy of
// half the bytes: by LDS-DMA for C_in = 16 / 32 (no staging registers: these variants are VGPR-limited and ran a
// load -> convert -> store loop with ONE load in flight per thread, a third of their time), by the register prefetch for 64.
template <int PRECISE, int CIN, int COUT, bool IN16 = false>
__global__ __launch_bounds__(256) void conv3d_igemm_kernel(const void* __restrict__ x_,
                                                           const typename Mma<PRECISE>::elem* __restrict__ wt,
                                                           float* __restrict__ y, float* __restrict__ stat_partials,
                                                           int n_bt, int T, int H, int W, int Ho, int Wo, int pad, int KP, int th) {
  using M = Mma<PRECISE>;
  static_assert(!IN16 || PRECISE != MODE_F32, "16-bit input needs a 16-bit MFMA mode");
  const float* x = reinterpret_cast<const float*>(x_);
  const unsigned short* x16 = reinterpret_cast<const unsigned short*>(x_);
  using E = typename M::elem;
  constexpr int ES = sizeof(E), EPC = 16 / ES;       // elements per 16-byte chunk
  // 16-bit input with 64 channels: the halo is staged in two halves of 32 channels (K order kd, half, tap, channel): 25.6 KB by LDS-DMA
  // like the 32-channel variants instead of 51 KB through a 100-register prefetch -- three to four workgroups per CU instead of two
  constexpr bool HSPLIT = PRECISE != MODE_F32 && CIN == 64;      // (also for f32 input: the two input forms stay bit-identical)
  constexpr int CH = HSPLIT ? 32 : CIN;               // channels per halo stage
  constexpr int NH = CIN / CH;                        // halo stages per kd plane
  constexpr int RBH = CH * ES, NCH = RBH / 16;        // halo: bytes / chunks per position
  constexpr int RBW = 64 * ES, NCW = RBW / 16;        // weight tile: bytes / chunks per row (64 k)
  constexpr int NT = COUT / 16;
  constexpr int NCHUNK = (25 * CH + 63) / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  E* halo = reinterpret_cast<E*>(smem);                         // [20*20][CH] swizzled
  // 16-bit modes (round 4): the weight tiles go global -> LDS by DMA into a RING OF THREE, two chunks ahead of their use, instead of through
  // registers one chunk ahead: at 56^2 / 28^2 these launches are bound by the 307 / 614 KB of weights every tile pulls through LDS, not by MFMAs
  constexpr bool WDMA = PRECISE != MODE_F32;
  constexpr int WSLOTS = WDMA ? 3 : 2;
  E* wl = halo + 400 * CH;                                      // [WSLOTS][COUT][64] swizzled
  float* red = reinterpret_cast<float*>(wl + WSLOTS * COUT * 64);    // [4][2][COUT] stats scratch

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l16 = lane & 15;
  const int ny = (Ho + th - 1) / th;
  const TileId tile = xcd_tile((Wo + 15) / 16, ny, (int64_t)((Wo + 15) / 16) * ny * n_bt);
  if (!tile.valid) return;
  const int x0 = tile.tx * 16, y0 = tile.ty * th;
  const int row0 = tile_row0(wv, th);
  const bool four = tile_nrows(wv, th) == 4;      // wave-uniform: the wave's fourth row exists
  const int hpos = (th + 4) * 20;                 // halo positions of a tile
  const int bt = tile.bt, t = bt % T;
  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- halo staging.  C_in = 64: the 25 float4 of a thread's share of the NEXT frame's halo are requested before this
  // frame's MFMAs and converted / written to LDS after them (register prefetch).  With a plain load -> convert -> store
  // loop one load is in flight per thread: 25 dependent round trips per frame, and these variants run only two
  // workgroups per CU (60 - 70 KB of LDS each), too few to cover that; the same LDS limit leaves 256 VGPRs per lane,
  // so the 100 staging registers are free.  (Scratch build without the halo loads: igemm 4.3 -> 2.9 ms per step.)
  // The smaller C_in variants run 3 - 5 workgroups per CU and keep the plain loop (batched loads cost them occupancy).
  constexpr int VE = IN16 ? 8 : 4;                      // elements per 16-byte global vector
  constexpr int HV = 400 * (CH / VE), NV = (HV + 255) / 256;
  constexpr bool PREFETCH = CIN >= 64 && !HSPLIT;
  float4 hv[PREFETCH ? NV : 1];                         // IN16: the same 16 bytes hold 8 operand elements
  auto fetch = [&](int kd) __attribute__((always_inline)) {
    const int64_t plane = (int64_t)(bt + kd - 1) * H * W * CIN;
    int tv = tid;
    asm volatile("" : "+v"(tv));   // the index arithmetic is redone per call: hoisted out of the kd loop it costs 100+ registers
#pragma unroll
    for (int j = 0; j < (PREFETCH ? NV : 1); ++j) {
      const int i = tv + j * 256;
      const int pos = i / (CIN / VE), cv = (i % (CIN / VE)) * VE;
      const int r = pos / 20, c = pos % 20;
      const int iy = y0 + r - pad, ix = x0 + c - pad;
      hv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < HV && iy >= 0 && iy < H && ix >= 0 && ix < W) {
        const int64_t e = plane + ((int64_t)iy * W + ix) * CIN + cv;
        if constexpr (IN16) hv[j] = *reinterpret_cast<const float4*>(x16 + e);
        else hv[j] = *reinterpret_cast<const float4*>(x + e);
      }
    }
  };
  auto stash = [&]() __attribute__((always_inline)) {
    int tv = tid;
    asm volatile("" : "+v"(tv));
#pragma unroll
    for (int j = 0; j < (PREFETCH ? NV : 1); ++j) {
      const int i = tv + j * 256;
      if (i < HV) {
        const int pos = i / (CIN / VE), cv = (i % (CIN / VE)) * VE, c = pos % 20;
        E* d = halo + (pos * NCH + swz_halo<RBH, ES>(c, cv / EPC)) * EPC + (cv % EPC);
        if constexpr (IN16) {
          *reinterpret_cast<float4*>(d) = hv[j];
        } else {
          d[0] = M::cvt(hv[j].x); d[1] = M::cvt(hv[j].y); d[2] = M::cvt(hv[j].z); d[3] = M::cvt(hv[j].w);
        }
      }
    }
  };
  const int kd_lo = t == 0 ? 1 : 0, kd_hi = t == T - 1 ? 1 : 2;   // frames t + kd - 1 inside the clip (block-uniform)
  if constexpr (PREFETCH) fetch(kd_lo);
  for (int kd = kd_lo; kd <= kd_hi; ++kd)
  for (int hh = 0; hh < NH; ++hh) {
    __syncthreads();
    // ---- stage the 20x20xCH halo of frame t + kd - 1 (zero-filled outside the image)
    if constexpr (PREFETCH) {
      stash();
    } else if constexpr (IN16) {
      // LDS-DMA: 16 B per lane straight into the halo image.  The DMA writes lane-linearly (slot i = position i / NCH, physical
      // chunk i % NCH), so the swizzle is applied on the SOURCE side (XOR: its own inverse).  Positions outside the image read
      // the zero padding at the end of weight row 0 (k >= 25 C_in: maavss_conv3d_kp leaves at least 64 bytes) -- no predication, the
      // wave stays whole, the destination base stays lane 0's.
      const unsigned short* xp = x16 + (int64_t)(bt + kd - 1) * H * W * CIN + hh * CH;
      const unsigned short* zeros = reinterpret_cast<const unsigned short*>(wt) + 25 * CIN;
      const int hv_rt = hpos * NCH;               // (th + 4) halo rows
      for (int i0 = 0; i0 < hv_rt; i0 += 256) {
        const int i = i0 + tid;
        if (i < hv_rt) {
          const int pos = i / NCH, pc = i % NCH;
          const int r = pos / 20, c = pos % 20;
          const int iy = y0 + r - pad, ix = x0 + c - pad;
          const unsigned short* src = zeros;
          if (iy >= 0 && iy < H && ix >= 0 && ix < W) src = xp + ((int64_t)iy * W + ix) * CIN + swz_halo<RBH, ES>(c, pc) * EPC;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(halo + (int64_t)i * EPC), 16, 0, 0);
        }
      }
    } else {
      const float* xp = x + (int64_t)(bt + kd - 1) * H * W * CIN + hh * CH;
      for (int i = tid; i < hpos * (CH / 4); i += 256) {
        const int pos = i / (CH / 4), c4 = (i % (CH / 4)) * 4;
        const int r = pos / 20, c = pos % 20;
        const int iy = y0 + r - pad, ix = x0 + c - pad;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const float4*>(xp + ((int64_t)iy * W + ix) * CIN + c4);
        E* d = halo + (pos * NCH + swz_halo<RBH, ES>(c, c4 / EPC)) * EPC + (c4 % EPC);
        d[0] = M::cvt(v.x); d[1] = M::cvt(v.y); d[2] = M::cvt(v.z); d[3] = M::cvt(v.w);
      }
    }
    // ---- weight chunks of this (kd, half).  Source of the 16-byte piece c of row n of chunk q: k = 64 q + 8 c, or with the halo in
    // halves k = (2 q + c / 4) * 64 + 32 half + 8 (c % 4) -- two taps x 32 channels; the phantom tap 25 of the last chunk is the zero tail
    const E* wk = wt + (int64_t)kd * COUT * KP;
    auto wsrc = [&](int i, int q) __attribute__((always_inline)) {
      const int n = i / NCW, c = i % NCW;
      if constexpr (HSPLIT) return wk + (int64_t)n * KP + (2 * q + c / 4) * 64 + hh * 32 + (c % 4) * EPC;
      else return wk + (int64_t)n * KP + q * 64 + c * EPC;
    };
    constexpr int WV = (COUT * NCW + 255) / 256;
    // DMA of chunk q into ring slot `slot`: lane-linear destination (piece i = row i / 8, physical chunk i % 8), swizzle on the source side
    auto wdma = [&](int q, int slot) __attribute__((always_inline)) {
#pragma unroll
      for (int v = 0; v < WV; ++v) {
        const int i = v * 256 + tid;
        if (WV * 256 == COUT * NCW || i < COUT * NCW) {
          const int n = i / NCW, pc = i % NCW;
          const E* src = wsrc(n * NCW + swz<RBW>(n, pc), q);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(wl + slot * COUT * 64 + (int64_t)i * EPC), 16, 0, 0);
        }
      }
    };
    if constexpr (WDMA) {
      wdma(0, 0);
      if (NCHUNK > 1) wdma(1, 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // halo (DMA) and the first two weight tiles have landed
      __syncthreads();
    } else {
      for (int i = tid; i < COUT * NCW; i += 256) {
        const int n = i / NCW, c = i % NCW;
        *reinterpret_cast<uint4*>(wl + (n * NCW + swz<RBW>(n, c)) * EPC) = *reinterpret_cast<const uint4*>(wsrc(i, 0));
      }
      if constexpr (IN16 && !PREFETCH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the halo DMA has landed
      __syncthreads();
    }
    if constexpr (PREFETCH) {
      if (kd < kd_hi) fetch(kd + 1);
    }
    int slot = 0;
    for (int ch = 0; ch < NCHUNK; ++ch) {
      uint4 wreg[WDMA ? 1 : WV];
      if constexpr (WDMA) {
        // slot of chunk ch + 2 = slot of chunk ch - 1: every wave is past the barrier that ended it
        if (ch + 2 < NCHUNK) wdma(ch + 2, slot == 0 ? 2 : slot - 1);
      } else {
        // unconditional (clamped) loads keep wreg in registers: a conditionally written array lands in scratch
        const int chn = ch + 1 < NCHUNK ? ch + 1 : ch;
#pragma unroll
        for (int v = 0; v < WV; ++v) {
          int i = v * 256 + tid;
          i = i < COUT * NCW ? i : 0;
          wreg[v] = *reinterpret_cast<const uint4*>(wsrc(i, chn));
        }
      }
      const E* wb = wl + (WDMA ? slot : (ch & 1)) * COUT * 64;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int kk = ch * 64 + s * 32 + 8 * g;
        int tap = kk / CH;
        const int ci = kk % CH;
        tap = tap > 24 ? 24 : tap;  // padded tail: weights are zero there
        const int kh = tap / 5, kw = tap % 5;
        typename M::frag fa[4], fb[NT];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = row0 + i + kh, c = l16 + kw;
          const E* base = halo + (r * 20 + c) * NCH * EPC;
          if constexpr (PRECISE == MODE_F32) {
            fa[i].lo = *reinterpret_cast<const f32x4*>(base + swz_halo<RBH, ES>(c, ci / EPC) * EPC);
            fa[i].hi = *reinterpret_cast<const f32x4*>(base + swz_halo<RBH, ES>(c, ci / EPC + 1) * EPC);
          } else {
            fa[i] = M::load(base + swz_halo<RBH, ES>(c, ci / EPC) * EPC);
          }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int n = j * 16 + l16;
          const E* base = wb + n * 64;
          const int c0 = (s * 32 + 8 * g) / EPC;
          if constexpr (PRECISE == MODE_F32) {
            fb[j].lo = *reinterpret_cast<const f32x4*>(base + swz<RBW>(n, c0) * EPC);
            fb[j].hi = *reinterpret_cast<const f32x4*>(base + swz<RBW>(n, c0 + 1) * EPC);
          } else {
            fb[j] = M::load(base + swz<RBW>(n, c0) * EPC);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < 3 || four) {
#pragma unroll
            for (int j = 0; j < NT; ++j) M::mma(acc[i][j], fa[i], fb[j]);
          }
      }
      if constexpr (WDMA) {
        // chunk ch + 1 must have landed (this wave's pieces; the barrier collects the others'), chunk ch + 2 may stay in flight
        if (ch + 2 < NCHUNK) {
          if constexpr (WV == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        slot = slot == 2 ? 0 : slot + 1;
      } else {
        if (ch + 1 < NCHUNK) {
          E* wn = wl + ((ch + 1) & 1) * COUT * 64;
#pragma unroll
          for (int v = 0; v < WV; ++v) {
            const int i = v * 256 + tid;
            if (i < COUT * NCW) {
              const int n = i / NCW, c = i % NCW;
              *reinterpret_cast<uint4*>(wn + (n * NCW + swz<RBW>(n, c)) * EPC) = wreg[v];
            }
          }
        }
        __syncthreads();
      }
    }
  }
  // ---- epilogue: store + optional per-block BatchNorm partial sums (sum, sum of squares per channel)
  float* yp = y + (int64_t)bt * Ho * Wo * COUT;
  float s1[NT], s2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int oy = y0 + row0 + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ox = x0 + g * 4 + r;
      if ((i < 3 || four) && oy < Ho && ox < Wo) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float v = acc[i][j][r];
          yp[((int64_t)oy * Wo + ox) * COUT + j * 16 + l16] = v;
          s1[j] += v;
          s2[j] += v * v;
        }
      }
    }
  }
  if (stat_partials != nullptr) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      s1[j] = rows4_sum(s1[j]);
      s2[j] = rows4_sum(s2[j]);
      if (g == 0) {
        red[(wv * 2 + 0) * COUT + j * 16 + l16] = s1[j];
        red[(wv * 2 + 1) * COUT + j * 16 + l16] = s2[j];
      }
    }
    __syncthreads();
    if (tid < 2 * COUT) {
      const float v = red[tid] + red[2 * COUT + tid] + red[4 * COUT + tid] + red[6 * COUT + tid];
      stat_partials[tile.lin * 2 * COUT + tid] = v;
    }
  }
}

template <int PRECISE, int CIN, int COUT, bool IN16 = false>
static int launch_igemm(const void* x, const void* wt, float* y, float* stats, int B, int T, int H, int W, int Ho,
                        int Wo, int pad, int KP, hipStream_t st) {
  using E = typename Mma<PRECISE>::elem;
  const int th = conv_tile_h(Ho);
  const size_t smem = (400 * (PRECISE != MODE_F32 && CIN == 64 ? 32 : CIN) + (PRECISE != MODE_F32 ? 3 : 2) * COUT * 64) * sizeof(E) + 8 * COUT * sizeof(float);
  auto kern = conv3d_igemm_kernel<PRECISE, CIN, COUT, IN16>;
  if (smem > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const int64_t tiles = (int64_t)cdiv(Wo, 16) * cdiv(Ho, th) * B * T;
  hipLaunchKernelGGL(kern, dim3(xcd_grid(tiles)), dim3(256), smem, st, x, reinterpret_cast<const E*>(wt), y, stats, B * T, T, H, W, Ho, Wo, pad, KP, th);
  return 0;
}

// f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}): a loop whose index is a compile-time constant in the body
template <class F, int... S>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, S...>) { (f(std::integral_constant<int, S>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// --------------------------------------------------------------------------------------------
// The 16-bit modes of the implicit GEMM (round 4): the tile, the K order (kd, 32-channel half of a 64-channel input, tap, channel), the halo
// image and the epilogue of conv3d_igemm_kernel, with the inner loop rebuilt around what its counters showed (35-45 % matrix-pipe busy, 3-4.5
// vector instructions per MFMA, waves waiting 45-64 % of their cycles; profiles/r3_e_kernel_pmc.json):
//   * the 13 / 25 K steps of a halo stage are unrolled with the tap as a compile-time constant: a lane keeps one swizzled byte offset per
//     kw (32-channel stages) or one offset + a per-step select (16-channel stages: a 32-deep step spans two taps), the row / tap part is the
//     ds_read's immediate -- no address arithmetic in the loop;
//   * the fragments of step t + 1 are read before the MFMAs of step t (two register sets);
//   * the weight chunks (64 k) go global -> LDS by DMA into a ring of four tiles, requested three chunks ahead, so the fragments of the next
//     chunk can be read before the barrier that ends this one and a request has two chunks of time to land; the barrier waits for the
//     request before the newest only.
template <int PRECISE, int CIN, int COUT, bool IN16>
__global__ __launch_bounds__(256) void conv3d_igemm16_kernel(const void* __restrict__ x_, const typename Mma<PRECISE>::elem* __restrict__ wt,
                                                             float* __restrict__ y, float* __restrict__ stat_partials, int n_bt, int T, int H,
                                                             int W, int Ho, int Wo, int pad, int KP, int th) {
  using M = Mma<PRECISE>;
  using E = typename M::elem;
  static_assert(PRECISE != MODE_F32 && sizeof(E) == 2, "16-bit MFMA modes only");
  constexpr int CH = CIN == 64 ? 32 : CIN;            // channels per halo stage
  constexpr int NH = CIN / CH;                        // halo stages per kd plane
  constexpr int PB = CH * 2;                          // bytes per halo position
  constexpr int NCH = PB / 16;                        // 16-byte chunks per halo position
  constexpr int NT = COUT / 16;
  constexpr int STEPS = (25 * CH + 31) / 32;          // 32-deep K steps per stage: 13 (two taps each, the last half phantom) or 25 (one tap each)
  constexpr int NCHUNK = (STEPS + 1) / 2;             // 64-k weight chunks per stage
  constexpr int WB = COUT * 128;                      // bytes of one weight tile [COUT][64]
  constexpr int WV = (COUT * 8 + 255) / 256;          // 16-byte pieces of a weight tile per thread
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;                                  // [400][PB], chunk-swizzled (swz_halo)
  constexpr int RING = 4;
  char* ring = smem + 400 * PB;                       // [RING][COUT][128 B], chunk ^= (n >> 1) & 7
  float* red = reinterpret_cast<float*>(ring + RING * WB);
  const float* x = reinterpret_cast<const float*>(x_);
  const unsigned short* x16 = reinterpret_cast<const unsigned short*>(x_);

  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l16 = lane & 15;
  const int ny = (Ho + th - 1) / th;
  const TileId tile = xcd_tile((Wo + 15) / 16, ny, (int64_t)((Wo + 15) / 16) * ny * n_bt);
  if (!tile.valid) return;
  const int x0 = tile.tx * 16, y0 = tile.ty * th;
  const int row0 = tile_row0(wv, th);
  const bool four = tile_nrows(wv, th) == 4;      // wave-uniform: the wave's fourth row exists
  const int hpos = (th + 4) * 20;                 // halo positions of a tile
  const int bt = tile.bt, t = bt % T;
  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- per-lane LDS byte offsets.  A (halo): position (row 4 wv + i + kh, column l16 + kw), chunk = the lane's 8 channels
  unsigned a_off[CH == 32 ? 5 : 1];
  if constexpr (CH == 32) {
#pragma unroll
    for (int kw = 0; kw < 5; ++kw) {
      const int c = l16 + kw;
      a_off[kw] = (unsigned)(((row0 * 20 + c) * NCH + swz_halo<PB, 2>(c, g)) * 16);
    }
  } else {
    a_off[0] = (unsigned)(((row0 * 20 + l16) * NCH + (g & 1)) * 16);      // + tap offset of the lane's half of the step
  }
  // B (weight tile): row n = 16 j + l16, chunk 4 s + g of the row, swizzled with (n >> 1) & 7 = (l16 >> 1) & 7
  unsigned b_off[2];
#pragma unroll
  for (int sp = 0; sp < 2; ++sp) b_off[sp] = (unsigned)((l16 * 8 + ((sp * 4 + g) ^ ((l16 >> 1) & 7))) * 16);

  typename M::frag fa[2][4], fb[2][NT];
  // fragments of step `st` (compile-time) of the current stage into register set `set`; `slot_b` = byte offset of the ring tile of its chunk
  auto load_frags = [&](auto st_c, int set, unsigned slot_b) __attribute__((always_inline)) {
    constexpr int st = decltype(st_c)::value;
    unsigned ab;
    int imm;                                           // compile-time part of the A address (rows / taps)
    if constexpr (CH == 32) {
      constexpr int kh = st / 5, kw = st % 5;
      ab = a_off[kw];
      imm = kh * 20 * PB;
    } else {
      constexpr int t0 = 2 * st, t1 = 2 * st + 1 > 24 ? 24 : 2 * st + 1;       // tap 25 is the zero tail of the weight rows
      constexpr int o0 = ((t0 / 5) * 20 + t0 % 5) * PB, o1 = ((t1 / 5) * 20 + t1 % 5) * PB;
      ab = a_off[0] + (g >= 2 ? (unsigned)o1 : (unsigned)o0);
      imm = 0;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[set][i] = *reinterpret_cast<const typename M::frag*>(halo + ab + imm + i * 20 * PB);
    const unsigned bb = slot_b + b_off[st & 1];
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[set][j] = *reinterpret_cast<const typename M::frag*>(ring + bb + j * 16 * 128);
  };
  auto mfmas = [&](int set) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < 3 || four) {       // (the fourth row's fragment is read either way: inside the 20-row halo allocation)
#pragma unroll
        for (int j = 0; j < NT; ++j) M::mma(acc[i][j], fa[set][i], fb[set][j]);
      }
  };

  // ---- weight tiles by LDS-DMA: piece i of a tile = 16 bytes, lane-linear destination (row n = i / 8, PHYSICAL chunk i % 8): the swizzle is
  // applied on the source side
  const int kd_lo = t == 0 ? 1 : 0, kd_hi = t == T - 1 ? 1 : 2;   // frames t + kd - 1 inside the clip (block-uniform)
  auto wdma = [&](const E* wk, int hh, int q, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int v = 0; v < WV; ++v) {
      const int i = v * 256 + tid;
      if (WV * 256 == COUT * 8 || i < COUT * 8) {
        const int n = i >> 3, c = (i & 7) ^ ((n >> 1) & 7);
        const E* src = CIN == 64 ? wk + (int64_t)n * KP + (2 * q + (c >> 2)) * 64 + hh * 32 + (c & 3) * 8 : wk + (int64_t)n * KP + q * 64 + c * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(ring + slot * WB + (int64_t)i * 16), 16, 0, 0);
      }
    }
  };

  for (int kd = kd_lo; kd <= kd_hi; ++kd)
  for (int hh = 0; hh < NH; ++hh) {
    // (every wave is past its last read of the previous stage's halo and ring: the barrier that ended its last chunk)
    // ---- halo of frame t + kd - 1, channels [hh CH, hh CH + CH), zero outside the image
    if constexpr (IN16) {
      // LDS-DMA, 16 B per lane, lane-linear destination: the swizzle is applied on the SOURCE side (XOR: its own inverse); positions outside
      // the image read the zero tail of weight row 0 (k >= 25 C_in: maavss_conv3d_kp leaves at least 64 bytes)
      const unsigned short* xp = x16 + (int64_t)(bt + kd - 1) * H * W * CIN + hh * CH;
      const unsigned short* zeros = reinterpret_cast<const unsigned short*>(wt) + 25 * CIN;
      int tv = tid;
      asm volatile("" : "+v"(tv));   // the index arithmetic is redone per stage: hoisted out of the stage loop it costs 60 registers
      for (int i0 = 0; i0 < hpos * NCH; i0 += 256) {
        const int i = i0 + tv;
        if (i < hpos * NCH) {
          const int pos = i / NCH, pc = i % NCH;
          const int r = pos / 20, c = pos % 20;
          const int iy = y0 + r - pad, ix = x0 + c - pad;
          const unsigned short* src = zeros;
          if (iy >= 0 && iy < H && ix >= 0 && ix < W) src = xp + ((int64_t)iy * W + ix) * CIN + swz_halo<PB, 2>(c, pc) * 8;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(halo + (int64_t)i * 16), 16, 0, 0);
        }
      }
    } else {
      const float* xp = x + (int64_t)(bt + kd - 1) * H * W * CIN + hh * CH;
      int tv = tid;
      asm volatile("" : "+v"(tv));
      for (int i = tv; i < hpos * (CH / 4); i += 256) {
        const int pos = i / (CH / 4), c4 = (i % (CH / 4)) * 4;
        const int r = pos / 20, c = pos % 20;
        const int iy = y0 + r - pad, ix = x0 + c - pad;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const float4*>(xp + ((int64_t)iy * W + ix) * CIN + c4);
        E* d = reinterpret_cast<E*>(halo + (pos * NCH + swz_halo<PB, 2>(c, c4 / 8)) * 16) + (c4 % 8);
        d[0] = M::cvt(v.x); d[1] = M::cvt(v.y); d[2] = M::cvt(v.z); d[3] = M::cvt(v.w);
      }
    }
    // ---- weight chunks 0 .. 2 of the stage into ring tiles 0 .. 2
    const E* wk = wt + (int64_t)kd * COUT * KP;
    wdma(wk, hh, 0, 0);
    if (NCHUNK > 1) wdma(wk, hh, 1, 1);
    if (NCHUNK > 2) wdma(wk, hh, 2, 2);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // halo (DMA or stores) and the three tiles have landed
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    load_frags(std::integral_constant<int, 0>{}, 0, 0u);
    __builtin_amdgcn_sched_barrier(0);
    // ---- the K steps of the stage.  Chunk c + 3 is requested when chunk c starts (its tile was chunk c - 1's: every wave is past the barrier
    // that ended it) and has to be visible when chunk c + 1 ends (the fragments of chunk c + 2's first step are read there): two chunks of time.
    auto step = [&](auto st_c) __attribute__((always_inline)) {
      constexpr int st = decltype(st_c)::value;
      constexpr int c = st / 2;                        // chunk of this step
      if constexpr ((st & 1) == 0 && c + 3 < NCHUNK) wdma(wk, hh, c + 3, (c + 3) % RING);
      if constexpr (st + 1 < STEPS) load_frags(std::integral_constant<int, st + 1>{}, (st + 1) & 1, (unsigned)((((st + 1) / 2) % RING) * WB));
      __builtin_amdgcn_sched_barrier(0);
      mfmas(st & 1);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((st & 1) || st + 1 == STEPS) {     // the chunk ends: chunk c + 2 has landed (chunk c + 3 may stay in flight)
        if constexpr (c + 3 < NCHUNK) {
          if constexpr (WV == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    };
    static_for<STEPS>(step);
  }
  // ---- epilogue: store + optional per-block BatchNorm partial sums (sum, sum of squares per channel)
  float* yp = y + (int64_t)bt * Ho * Wo * COUT;
  float s1[NT], s2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) s1[j] = s2[j] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int oy = y0 + row0 + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ox = x0 + g * 4 + r;
      if ((i < 3 || four) && oy < Ho && ox < Wo) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float v = acc[i][j][r];
          yp[((int64_t)oy * Wo + ox) * COUT + j * 16 + l16] = v;
          s1[j] += v;
          s2[j] += v * v;
        }
      }
    }
  }
  if (stat_partials != nullptr) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      s1[j] = rows4_sum(s1[j]);
      s2[j] = rows4_sum(s2[j]);
      if (g == 0) {
        red[(wv * 2 + 0) * COUT + j * 16 + l16] = s1[j];
        red[(wv * 2 + 1) * COUT + j * 16 + l16] = s2[j];
      }
    }
    __syncthreads();
    if (tid < 2 * COUT) {
      const float v = red[tid] + red[2 * COUT + tid] + red[4 * COUT + tid] + red[6 * COUT + tid];
      stat_partials[tile.lin * 2 * COUT + tid] = v;
    }
  }
}

template <int PRECISE, int CIN, int COUT, bool IN16>
static int launch_igemm16(const void* x, const void* wt, float* y, float* stats, int B, int T, int H, int W, int Ho, int Wo, int pad, int KP,
                          hipStream_t st) {
  {
    using E = typename Mma<PRECISE>::elem;
    const size_t smem = 400 * (CIN == 64 ? 32 : CIN) * 2 + 4 * COUT * 128 + 8 * COUT * sizeof(float);
    auto kern = conv3d_igemm16_kernel<PRECISE, CIN, COUT, IN16>;
    if (smem > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const int th = conv_tile_h(Ho);
    const int64_t tiles = (int64_t)cdiv(Wo, 16) * cdiv(Ho, th) * B * T;
    hipLaunchKernelGGL(kern, dim3(xcd_grid(tiles)), dim3(256), smem, st, x, reinterpret_cast<const E*>(wt), y, stats, B * T, T, H, W, Ho, Wo, pad, KP, th);
    return 0;
  }
}

// padded K of a weight row: a multiple of 64 that leaves at least 32 zero elements (64 bytes) behind the 25 C_in real ones -- the zero source of
// the halo LDS-DMA and the phantom tap of the split-halo variants
extern "C" int maavss_conv3d_kp(int c_in) { return ((25 * c_in + 32 + 63) / 64) * 64; }

extern "C" int maavss_conv3d_prep_weights(const float* w, void* wt, int c_out, int c_in, int mode, int precise, void* stream) {
  MAAVSS_CHECK_ARG(w && wt, "conv3d_prep_weights: null pointer");
  const int cin = mode ? c_out : c_in, nN = mode ? c_in : c_out;
  const int KP = maavss_conv3d_kp(cin);
  const int64_t total = 3LL * nN * KP;
  dim3 grid(min(1024, cdiv(total, 256)));
  MAAVSS_CHECK_ARG(precise >= 0 && precise <= 2, "conv3d_prep_weights: mode must be 0 (bf16), 1 (f32) or 2 (f16)");
  if (precise == MODE_F32) hipLaunchKernelGGL(conv3d_prep_w_kernel<MODE_F32>, grid, dim3(256), 0, (hipStream_t)stream, w, (float*)wt, c_out, c_in, KP, mode);
  else if (precise == MODE_F16) hipLaunchKernelGGL(conv3d_prep_w_kernel<MODE_F16>, grid, dim3(256), 0, (hipStream_t)stream, w, (unsigned short*)wt, c_out, c_in, KP, mode);
  else hipLaunchKernelGGL(conv3d_prep_w_kernel<MODE_BF16>, grid, dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)wt, c_out, c_in, KP, mode);
  MAAVSS_LAUNCH_CHECK("conv3d_prep_w_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_conv3d_igemm(const void* x, const void* wt, float* y, float* stat_partials, int B, int T, int H,
                                   int W, int c_in, int c_out, int pad, int precise, int x16, void* stream) {
  MAAVSS_CHECK_ARG(x && wt && y, "conv3d_igemm: null pointer");
  MAAVSS_CHECK_ARG(!x16 || precise != MODE_F32, "conv3d_igemm: 16-bit input needs precise = 0 (bf16) or 2 (f16)");
  MAAVSS_CHECK_ARG(pad >= 0 && pad <= 4, "conv3d_igemm: pad must be in [0,4]");
  MAAVSS_CHECK_ARG(precise >= 0 && precise <= 2, "conv3d_igemm: mode must be 0 (bf16), 1 (f32) or 2 (f16)");
  const int Ho = H + 2 * pad - 4, Wo = W + 2 * pad - 4;
  MAAVSS_CHECK_ARG(Ho > 0 && Wo > 0 && B > 0 && T > 0, "conv3d_igemm: empty output");
  MAAVSS_CHECK_ARG((int64_t)cdiv(Wo, 16) * cdiv(Ho, 16) * B * T < (1LL << 31) - 8, "conv3d_igemm: too many output tiles");
  const int KP = maavss_conv3d_kp(c_in);
  hipStream_t st = (hipStream_t)stream;
  // 16-bit modes: the pipelined kernel (conv3d_igemm16_kernel); MAAVSS_IGEMM_OLD=1 = the round-1..3 loop, kept as the measured baseline
  // (profiles/r4_igemm_bench.txt) and as the exact-f32 path
  static const bool old16 = getenv("MAAVSS_IGEMM_OLD") != nullptr;
#define CASE(CI, CO)                                                                                          \
  if (c_in == CI && c_out == CO) {                                                                            \
    if (precise == MODE_F32) launch_igemm<MODE_F32, CI, CO>(x, wt, y, stat_partials, B, T, H, W, Ho, Wo, pad, KP, st);      \
    else if (old16 && precise == MODE_F16 && x16) launch_igemm<MODE_F16, CI, CO, true>(x, wt, y, stat_partials, B, T, H, W, Ho, Wo, pad, KP, st); \
    else if (old16 && precise == MODE_F16) launch_igemm<MODE_F16, CI, CO>(x, wt, y, stat_partials, B, T, H, W, Ho, Wo, pad, KP, st); \
    else if (old16 && x16) launch_igemm<MODE_BF16, CI, CO, true>(x, wt, y, stat_partials, B, T, H, W, Ho, Wo, pad, KP, st);         \
    else if (old16) launch_igemm<MODE_BF16, CI, CO>(x, wt, y, stat_partials, B, T, H, W, Ho, Wo, pad, KP, st);                        \
    else if (precise == MODE_F16 && x16) launch_igemm16<MODE_F16, CI, CO, true>(x, wt, y, stat_partials, B, T, H, W, Ho, Wo, pad, KP, st); \
    else if (precise == MODE_F16) launch_igemm16<MODE_F16, CI, CO, false>(x, wt, y, stat_partials, B, T, H, W, Ho, Wo, pad, KP, st); \
    else if (x16) launch_igemm16<MODE_BF16, CI, CO, true>(x, wt, y, stat_partials, B, T, H, W, Ho, Wo, pad, KP, st);         \
    else launch_igemm16<MODE_BF16, CI, CO, false>(x, wt, y, stat_partials, B, T, H, W, Ho, Wo, pad, KP, st);                        \
    MAAVSS_LAUNCH_CHECK("conv3d_igemm_kernel");                                                               \
    return MAAVSS_OK;                                                                                         \
  }
  CASE(16, 32) CASE(32, 64) CASE(64, 64) CASE(64, 16) CASE(32, 16) CASE(64, 32) CASE(16, 64)
#undef CASE
  maavss_set_error("conv3d_igemm: unsupported channels %d -> %d", c_in, c_out);
  return MAAVSS_ERR_ARG;
}

// --------------------------------------------------------------------------------------------
// weight gradient
// DY16: dy arrives already rounded to the MFMA operand format (bn_pool_act_bwd writes it as bf16): copied, not converted
template <int PRECISE, int CI, int CO, bool DY16 = false>
__global__ __launch_bounds__(256) void conv3d_wgrad_kernel(const float* __restrict__ x, const void* __restrict__ dy_,
                                                           float* __restrict__ partials, int BT, int T, int H, int W,
                                                           int Ho, int Wo, int pad, int tiles_x, int tiles_y,
                                                           int tiles_per_chunk, int nchunk) {
  using M = Mma<PRECISE>;
  using E = typename M::elem;
  constexpr int MT = CI / 16, NT = CO / 16, NPAIR = 5 * MT, PW = (NPAIR + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  E* xs = reinterpret_cast<E*>(smem);  // [16 rows][20 cols][CI]
  E* ds = xs + 16 * 20 * CI;           // [16][16][CO]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int G = lane >> 4, l16 = lane & 15;
  // XCD-aware mapping: the 15 (kd,kh) blocks of one chunk walk the SAME x / dy tiles; give them 15 consecutive
  // slots of one XCD so that 14 of the 15 reads hit that XCD's L2 (measured before: 14.3 GB fetched per launch).
  // Each XCD owns a contiguous eighth of the chunks, so chunks that re-read each other's frames (kd planes) share an L2.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int chunk = xcd * ((nchunk + 7) / 8) + slot / 15, tg = slot % 15;
  if (chunk >= nchunk) return;
  const int kd = tg / 5, kh = tg % 5;
  f32x4 acc[PW][NT];
#pragma unroll
  for (int p = 0; p < PW; ++p)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[p][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int tiles_total = BT * tiles_x * tiles_y;
  const int tile_beg = chunk * tiles_per_chunk;
  const int tile_end = min(tiles_total, tile_beg + tiles_per_chunk);
  for (int tile = tile_beg; tile < tile_end; ++tile) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, bt = tile / (tiles_x * tiles_y);
    const int t = bt % T, tt = t + kd - 1;
    if (tt < 0 || tt >= T) continue;  // block-uniform
    const int x0 = tx * 16, y0 = ty * 16;
    __syncthreads();
    const float* xp = x + (int64_t)(bt + kd - 1) * H * W * CI;
    for (int i = tid; i < 320 * (CI / 4); i += 256) {
      const int pos = i / (CI / 4), c4 = (i % (CI / 4)) * 4;
      const int r = pos / 20, c = pos % 20;
      const int iy = y0 + r + kh - pad, ix = x0 + c - pad;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const float4*>(xp + ((int64_t)iy * W + ix) * CI + c4);
      E* d = xs + pos * CI + c4;
      d[0] = M::cvt(v.x); d[1] = M::cvt(v.y); d[2] = M::cvt(v.z); d[3] = M::cvt(v.w);
    }
    if constexpr (DY16) {
      static_assert(PRECISE != MODE_F32, "16-bit dy needs a 16-bit MFMA mode");
      const unsigned short* dp = reinterpret_cast<const unsigned short*>(dy_) + (int64_t)bt * Ho * Wo * CO;
      for (int i = tid; i < 256 * (CO / 8); i += 256) {
        const int pos = i / (CO / 8), c8 = (i % (CO / 8)) * 8;
        const int oy = y0 + pos / 16, ox = x0 + pos % 16;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (oy < Ho && ox < Wo) v = *reinterpret_cast<const uint4*>(dp + ((int64_t)oy * Wo + ox) * CO + c8);
        *reinterpret_cast<uint4*>(ds + pos * CO + c8) = v;
      }
    } else {
      const float* dp = reinterpret_cast<const float*>(dy_) + (int64_t)bt * Ho * Wo * CO;
      for (int i = tid; i < 256 * (CO / 4); i += 256) {
        const int pos = i / (CO / 4), c4 = (i % (CO / 4)) * 4;
        const int oy = y0 + pos / 16, ox = x0 + pos % 16;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (oy < Ho && ox < Wo) v = *reinterpret_cast<const float4*>(dp + ((int64_t)oy * Wo + ox) * CO + c4);
        E* d = ds + pos * CO + c4;
        d[0] = M::cvt(v.x); d[1] = M::cvt(v.y); d[2] = M::cvt(v.z); d[3] = M::cvt(v.w);
      }
    }
    __syncthreads();
#pragma unroll 2
    for (int ks = 0; ks < 8; ++ks) {
      // K step = output rows 2ks, 2ks+1; k = 0..31 -> (row 2ks + k/16, col k%16)
      typename M::frag fb[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if constexpr (PRECISE == MODE_F32) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int k = 8 * G + e;
            const float v = ds[((2 * ks + (k >> 4)) * 16 + (k & 15)) * CO + j * 16 + l16];
            if (e < 4) fb[j].lo[e] = v; else fb[j].hi[e - 4] = v;
          }
        } else {
          bf16x4 h[2];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const int k = 8 * G + 4 * hh + (l16 >> 2);
            const E* a = ds + ((2 * ks + (k >> 4)) * 16 + (k & 15)) * CO + j * 16 + (l16 & 3) * 4;
            h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a));
          }
          fb[j] = concat4(h[0], h[1]);
        }
      }
#pragma unroll
      for (int p = 0; p < PW; ++p) {
        const int q = wv + 4 * p;
        if (q < NPAIR) {  // wave-uniform
          const int kw = q / MT, mi = q % MT;
          typename M::frag fa;
          if constexpr (PRECISE == MODE_F32) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int k = 8 * G + e;
              const float v = xs[((2 * ks + (k >> 4)) * 20 + (k & 15) + kw) * CI + mi * 16 + l16];
              if (e < 4) fa.lo[e] = v; else fa.hi[e - 4] = v;
            }
          } else {
            bf16x4 h[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              const int k = 8 * G + 4 * hh + (l16 >> 2);
              const E* a = xs + ((2 * ks + (k >> 4)) * 20 + (k & 15) + kw) * CI + mi * 16 + (l16 & 3) * 4;
              h[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(a));
            }
            fa = concat4(h[0], h[1]);
          }
#pragma unroll
          for (int j = 0; j < NT; ++j) M::mma(acc[p][j], fa, fb[j]);
        }
      }
    }
  }
  // partials[chunk][tg][kw][ci][co]
  float* out = partials + ((int64_t)chunk * 15 + tg) * 5 * CI * CO;
#pragma unroll
  for (int p = 0; p < PW; ++p) {
    const int q = wv + 4 * p;
    if (q < NPAIR) {
      const int kw = q / MT, mi = q % MT;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[((int64_t)kw * CI + mi * 16 + G * 4 + r) * CO + j * 16 + l16] = acc[p][j][r];
    }
  }
}

// dW[co][ci][kd][kh][kw] (+)= sum_chunk partials[chunk][kd*5+kh][kw][ci][co]
// Sum of the per-chunk partials: 16 outputs x 16 chunk phases per 256-thread block (one thread per output walking all
// chunks serially took 70-260 us per layer).
__device__ __forceinline__ float chunk_sum16(const float* __restrict__ partials, int64_t total, int nchunk, int i0, bool& owner, int& i) {
  __shared__ float red[16][17];
  const int o = threadIdx.x & 15, ph = threadIdx.x >> 4;
  i = i0 + o;
  float s = 0.f;
  if (i < total)
    for (int c = ph; c < nchunk; c += 16) s += partials[(int64_t)c * total + i];
  red[ph][o] = s;
  __syncthreads();
  owner = ph == 0 && i < total;
  float t = 0.f;
  if (owner)
#pragma unroll
    for (int p = 0; p < 16; ++p) t += red[p][o];
  return t;
}

__global__ __launch_bounds__(256) void conv3d_wgrad_reduce_kernel(const float* __restrict__ partials, float* __restrict__ dw, int nchunk, int CI,
                                           int CO, int beta) {
  const int total = 75 * CI * CO;
  bool owner;
  int i;
  const float s = chunk_sum16(partials, total, nchunk, blockIdx.x * 16, owner, i);
  if (owner) {
    const int co = i % CO, ci = (i / CO) % CI, tap = i / (CO * CI);  // tap = (kd*5+kh)*5+kw
    float* d = dw + ((int64_t)co * CI + ci) * 75 + tap;
    *d = beta ? *d + s : s;
  }
}

int maavss_conv3d_wgrad_wide_try(const float* x, const void* dy, float* ws, int nchunk, int B, int T, int H, int W, int Ho,
                                 int Wo, int c_in, int c_out, int pad, int mode, int dy16, int x16, hipStream_t st);  // conv3d_wgrad_wide.hip

extern "C" int64_t maavss_conv3d_wgrad_ws_bytes(int c_in, int c_out, int nchunk) {
  return (int64_t)nchunk * 75 * c_in * c_out * 4;
}

template <int PRECISE, int CI, int CO, bool DY16 = false>
static void launch_wgrad(const float* x, const void* dy, float* ws, int BT, int T, int H, int W, int Ho, int Wo, int pad,
                         int nchunk, hipStream_t st) {
  using E = typename Mma<PRECISE>::elem;
  const size_t smem = (320 * CI + 256 * CO) * sizeof(E);
  auto kern = conv3d_wgrad_kernel<PRECISE, CI, CO, DY16>;
  if (smem > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const int tiles_x = cdiv(Wo, 16), tiles_y = cdiv(Ho, 16);
  const int tiles_total = BT * tiles_x * tiles_y;
  const int tpc = cdiv(tiles_total, nchunk);
  hipLaunchKernelGGL(kern, dim3(15 * cdiv(nchunk, 8) * 8), dim3(256), smem, st, x, dy, ws, BT, T, H, W, Ho, Wo, pad, tiles_x,
                     tiles_y, tpc, nchunk);
}

extern "C" int maavss_conv3d_wgrad(const void* x_, const void* dy, float* dw, float* ws, int nchunk, int B, int T, int H,
                                   int W, int c_in, int c_out, int pad, int beta, int precise, int in16, void* stream) {
  const float* x = reinterpret_cast<const float*>(x_);
  const int dy16 = in16 & 1, x16 = (in16 >> 1) & 1;      // bit 0: dy is bf16, bit 1: x is bf16 too
  MAAVSS_CHECK_ARG(x && dy && dw && ws, "conv3d_wgrad: null pointer");
  MAAVSS_CHECK_ARG(in16 >= 0 && in16 <= 3, "conv3d_wgrad: in16 is a 2-bit mask (1: dy bf16, 2: x bf16)");
  MAAVSS_CHECK_ARG(!dy16 || precise == MODE_BF16, "conv3d_wgrad: a 16-bit dy is bf16 and needs precise = 0");
  MAAVSS_CHECK_ARG(!x16 || (dy16 && ((c_in == 16 && c_out == 32) || (c_in == 32 && c_out == 64) || (c_in == 64 && c_out == 64))),
                   "conv3d_wgrad: a bf16 x needs a bf16 dy and one of the shapes 16->32, 32->64, 64->64 (got %d->%d)", c_in, c_out);
  MAAVSS_CHECK_ARG(nchunk >= 1, "conv3d_wgrad: nchunk must be >= 1");
  MAAVSS_CHECK_ARG(precise >= 0 && precise <= 2, "conv3d_wgrad: mode must be 0 (bf16), 1 (f32) or 2 (f16)");
  const int Ho = H + 2 * pad - 4, Wo = W + 2 * pad - 4;
  MAAVSS_CHECK_ARG(Ho > 0 && Wo > 0 && B > 0 && T > 0, "conv3d_wgrad: empty output");
  hipStream_t st = (hipStream_t)stream;
  // the two large-M layers use the wide kernel (conv3d_wgrad_wide.hip): every tile staged once / three times
  if (maavss_conv3d_wgrad_wide_try(x, dy, ws, nchunk, B, T, H, W, Ho, Wo, c_in, c_out, pad, precise, dy16, x16, st)) {
    MAAVSS_LAUNCH_CHECK("conv3d_wgrad_wide_kernel");
    hipLaunchKernelGGL(conv3d_wgrad_reduce_kernel, dim3(cdiv(75 * c_in * c_out, 16)), dim3(256), 0, st, ws, dw, nchunk, c_in,
                       c_out, beta);
    MAAVSS_LAUNCH_CHECK("conv3d_wgrad_reduce_kernel");
    return MAAVSS_OK;
  }
#define CASE(CI, CO)                                                                         \
  if (c_in == CI && c_out == CO) {                                                           \
    if (precise == MODE_F32) launch_wgrad<MODE_F32, CI, CO>(x, dy, ws, B * T, T, H, W, Ho, Wo, pad, nchunk, st);      \
    else if (precise == MODE_F16) launch_wgrad<MODE_F16, CI, CO>(x, dy, ws, B * T, T, H, W, Ho, Wo, pad, nchunk, st); \
    else if (dy16) launch_wgrad<MODE_BF16, CI, CO, true>(x, dy, ws, B * T, T, H, W, Ho, Wo, pad, nchunk, st);        \
    else launch_wgrad<MODE_BF16, CI, CO>(x, dy, ws, B * T, T, H, W, Ho, Wo, pad, nchunk, st);                        \
    MAAVSS_LAUNCH_CHECK("conv3d_wgrad_kernel");                                              \
    hipLaunchKernelGGL(conv3d_wgrad_reduce_kernel, dim3(cdiv(75 * CI * CO, 16)), dim3(256), 0, st, ws, dw, nchunk, CI, CO, beta); \
    MAAVSS_LAUNCH_CHECK("conv3d_wgrad_reduce_kernel");                                       \
    return MAAVSS_OK;                                                                        \
  }
  CASE(16, 32) CASE(32, 64) CASE(64, 64) CASE(64, 16)
#undef CASE
  maavss_set_error("conv3d_wgrad: unsupported channels %d -> %d", c_in, c_out);
  return MAAVSS_ERR_ARG;
}

// --------------------------------------------------------------------------------------------
// C_in = 1 (first layer): direct convolution.  x [BT][H][W], w16 [75][16] (tap-major), y [BT][H][W][16].
__global__ __launch_bounds__(256) void conv3d_c1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w16,
                                                            float* __restrict__ y, float* __restrict__ stat_partials,
                                                            int n_bt, int T, int H, int W) {
  __shared__ float halo[3][20][21];
  __shared__ float red[4][2][16];
  const int tid = threadIdx.x;
  const TileId tile = xcd_tile((W + 15) / 16, (H + 15) / 16, (int64_t)((W + 15) / 16) * ((H + 15) / 16) * n_bt);
  if (!tile.valid) return;
  const int x0 = tile.tx * 16, y0 = tile.ty * 16, bt = tile.bt, t = bt % T;
  for (int i = tid; i < 1200; i += 256) {
    const int kd = i / 400, r = (i % 400) / 20, c = i % 20;
    const int tt = t + kd - 1, iy = y0 + r - 2, ix = x0 + c - 2;
    float v = 0.f;
    if (tt >= 0 && tt < T && iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[((int64_t)(bt + kd - 1) * H + iy) * W + ix];
    halo[kd][r][c] = v;
  }
  __syncthreads();
  const int ly = tid >> 4, lx = tid & 15;
  float acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = 0.f;
#pragma unroll
  for (int kd = 0; kd < 3; ++kd)
#pragma unroll
    for (int kh = 0; kh < 5; ++kh)
#pragma unroll
      for (int kw = 0; kw < 5; ++kw) {
        const float v = halo[kd][ly + kh][lx + kw];
        const float* wp = w16 + ((kd * 5 + kh) * 5 + kw) * 16;  // wave-uniform -> scalar loads
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = fmaf(v, wp[c], acc[c]);
      }
  const int oy = y0 + ly, ox = x0 + lx;
  const bool ok = oy < H && ox < W;
  if (ok) {
    float4* o = reinterpret_cast<float4*>(y + (((int64_t)bt * H + oy) * W + ox) * 16);
    o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    o[2] = make_float4(acc[8], acc[9], acc[10], acc[11]);
    o[3] = make_float4(acc[12], acc[13], acc[14], acc[15]);
  }
  if (stat_partials != nullptr) {
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float v = ok ? acc[c] : 0.f;
      const float s1 = wave_sum(v), s2 = wave_sum(v * v);
      if (lane == 0) { red[wv][0][c] = s1; red[wv][1][c] = s2; }
    }
    __syncthreads();
    if (tid < 32) {
      const int which = tid >> 4, c = tid & 15;
      const float v = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
      stat_partials[tile.lin * 32 + tid] = v;
    }
  }
}

// The same layer on the matrix pipe (16-bit path): implicit GEMM  y[pos][co] = sum_k A[pos][k] W[k][co].  One workgroup = a 16x16
// output tile of one (b, t) plane (as the f32 kernel); a wave owns 4 rows of 16 positions = 4 M-tiles.
// K layout (round 3, second form): the 15 (kd, kh) tap rows are the 16-lane K groups, each 8 wide: kw = 0..4 + three zero-weight
// slots, K = 128 = four 32-deep MFMA steps, group G = 4 m + g <-> (kd, kh) = (G / 5, G % 5), G = 15 all zero.  A lane's fragment is
// then 8 CONSECUTIVE halo columns l16 .. l16 + 7 of one row: 16 bytes.  To make that one aligned LDS read the halo is kept as FOUR
// copies shifted by 0..3 columns (copy s [kd][row][j] = halo[kd][row][j + s], 20 columns, 2400 B each -- a stride that puts the four
// copies 32 B apart modulo the 128-B bank cycle): lane l16 = 4 q + s reads copy s at column 4 q, 8-byte aligned, as one ds_read2_b64.
// 16 LDS reads + 16 MFMAs per wave and tile.  The first form (k = kh * 5 + kw over one 32-deep step per kd plane) built each fragment
// from eight ds_read_u16 and four packs: 96 LDS instructions + 52 VALU per wave and tile, and with the per-tile 64-bit index
// divisions on the CU's one scalar unit the statistics pass alone took 355 us; see DESIGN.md 9.
// A workgroup walks C1_TPW consecutive tiles (tx fastest): weight fragments and addresses are set up once, the next tile's halo is
// requested before this tile's MFMAs and written to the other LDS image after them, BatchNorm partial sums are reduced once per
// workgroup (one row of `stat_partials` per workgroup).
//
// The conv output of this layer (1.6 GB at 32 x 16 x 224^2, the largest tensor of the step) does not have to exist.  The layer is
// 59 GFLOP on a matrix pipe that is idle here and its input is 103 MB -- so the 16-bit path runs the convolution THREE times instead
// of writing it once and reading it twice:
//   EPI 1  (conv3d_c1_stats)        conv -> BatchNorm partial sums only (the store happens only when a channel's |gamma| is below
//                                   BN_INV_MIN_GAMMA: the backward reduction then has to gather xhat from y, bn_pool.hip);
//   EPI 2  (conv3d_c1_bn_pool_act)  conv again -> gamma (y - mean) invstd + beta -> 2x2 max pool -> LeakyReLU: the pooled
//                                   activation (f32 + IEEE half) and the argmax byte, bit-identical to bn_pool_act_fwd_kernel on
//                                   the stored y (same expressions, same scan order dy, dx, first maximum wins, NaN sticks);
//   conv3d_c1_wgrad_recompute_kernel   conv a third time for xhat at every position of the BatchNorm backward.
// EPI 0 is the storing form (eval-mode forward, tests).
#define C1_TPW 8
#define C1_MIN_GAMMA 1e-2f        // == BN_INV_MIN_GAMMA (bn_pool.hip): below it xhat is not recoverable from the pooled output
#define C1H_COPY 1200             // halves per shifted copy [3][20][20]
#define C1H_IMG (4 * C1H_COPY)    // halves per halo image (four copies): 9600 B (+ 16 B: the dump slot of C1Halo::stash)
#define C1H_IMG_ALLOC (C1H_IMG + 8)

// the five halo elements of a thread (element i = tid + 256 j of the [3][20][20] halo): tile-independent constants
struct C1Halo {
  int off[5];                     // offset inside the [bt][H][W] frame stack relative to (bt, y0, x0)
  int lds[5];                     // half index inside copy 0 = (kd * 20 + row) * 20 + column
  __device__ __forceinline__ void setup(int tid, int H, int W) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int i = tid + j * 256;
      const int d = i / 400, rr = (i % 400) / 20, cc = i % 20;
      off[j] = ((d - 1) * H + (rr - 2)) * W + (cc - 2);
      lds[j] = i;
    }
  }
  // values of the tile at (bt, t, y0, x0): zero outside the clip / the frame
  __device__ __forceinline__ void fetch(const float* __restrict__ x, int tid, int bt, int t, int y0, int x0, int T, int H, int W, float v[5]) const {
    const float* base = x + ((int64_t)bt * H + y0) * W + x0;
    const bool inner = t >= 1 && t + 1 < T && y0 >= 2 && y0 + 18 <= H && x0 >= 2 && x0 + 18 <= W;      // uniform
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      v[j] = 0.f;
      if (j < 4 || tid < 1200 - 1024) {
        if (inner) {
          v[j] = base[off[j]];
        } else {
          const int i = lds[j], d = i / 400, rr = (i % 400) / 20, cc = i % 20;
          const int tt = t + d - 1, iy = y0 + rr - 2, ix = x0 + cc - 2;
          if (tt >= 0 && tt < T && iy >= 0 && iy < H && ix >= 0 && ix < W) v[j] = base[off[j]];
        }
      }
    }
  }
  // write the 16-bit values into the four shifted copies of one image
  __device__ __forceinline__ void stash(unsigned short* img, int tid, const unsigned short h[5]) const {
#pragma unroll
    for (int j = 0; j < 5; ++j)
      if (j < 4 || tid < 1200 - 1024) {
        const int cc = (tid + j * 256) % 20;
#pragma unroll
        for (int sft = 0; sft < 4; ++sft)      // column cc of the halo is column cc - sft of copy sft; the first sft columns go to a dump slot (no branch)
          img[cc >= sft ? sft * (C1H_COPY - 1) + lds[j] : C1H_IMG] = h[j];
      }
  }
};

// forward operands of a lane (co / position column l16 = 4 q + s, K group g, wave wv): weight fragments of the four K steps and the
// byte offsets of its halo fragments (tile row 4 wv + i: add 40 i)
struct C1Conv {
  bf16x8 fw[4];
  unsigned ra[4];
  __device__ __forceinline__ void setup(const float* __restrict__ w, int l16, int g, int wv) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int G = 4 * m + g, Gc = G < 15 ? G : 14, kd = Gc / 5, kh = Gc % 5;
      const float* wp = w + l16 * 75 + kd * 25 + kh * 5;
      const bool live = G < 15;
      const float w0 = live ? wp[0] : 0.f, w1 = live ? wp[1] : 0.f, w2 = live ? wp[2] : 0.f, w3 = live ? wp[3] : 0.f, w4 = live ? wp[4] : 0.f;
      fw[m] = __builtin_bit_cast(bf16x8, make_uint4(pack2<MODE_F16>(w0, w1), pack2<MODE_F16>(w2, w3), pack2<MODE_F16>(w4, 0.f), 0u));
      ra[m] = (unsigned)(((l16 & 3) * C1H_COPY + (kd * 20 + 4 * wv + kh) * 20 + (l16 & ~3)) * 2);
    }
  }
  // acc[i][r] += y(channel 4 g + r, position (row 4 wv + i, column l16)) of the tile whose IEEE-half image is `img`
  __device__ __forceinline__ void tile(const unsigned short* img, f32x4 acc[4]) const {
    const char* hb = reinterpret_cast<const char*>(img);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint2 lo = *reinterpret_cast<const uint2*>(hb + ra[m] + i * 40), hi = *reinterpret_cast<const uint2*>(hb + ra[m] + i * 40 + 8);
        Mma<MODE_F16>::mma(acc[i], fw[m], __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y)));   // D[channel][position]
      }
  }
};

struct C1EpiArgs {
  const float* mean;
  const float* invstd;
  const float* gamma;             // EPI 1: decides the conditional store; EPI 2: the affine part
  const float* beta;
  float* out;                     // [BT][Hp][Wp][16] pooled activation
  unsigned short* out16;          // the same as IEEE half (next conv's operand), may be null
  unsigned short* out_bf16;       // the same as bf16 (the next conv's weight-gradient operand), may be null
  unsigned char* argmax;          // [BT][Hp][Wp][16] window position dy * 2 + dx of the maximum
  int Hp, Wp;
};
template <int EPI>
__global__ __launch_bounds__(256) void conv3d_c1_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 float* __restrict__ y, float* __restrict__ stat_partials,
                                                                 int n_bt, int T, int H, int W, C1EpiArgs ep) {
  __shared__ __attribute__((aligned(16))) unsigned short halo[2][C1H_IMG_ALLOC];
  __shared__ float red[4][2][16];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l16 = lane & 15, g = lane >> 4;
  // tile list in 32-bit arithmetic (the entry points check nx * ny * n_bt < 2^31) and advanced one tile at a time: 64-bit
  // divisions per tile run on the one scalar unit of a CU, shared by its 16 waves
  const unsigned nx = (W + 15) / 16, ny = (H + 15) / 16;
  const unsigned total = nx * ny * (unsigned)n_bt, nsuper = (total + C1_TPW - 1) / C1_TPW;
  const unsigned per = (nsuper + 7) / 8;
  const unsigned sup = (blockIdx.x & 7) * per + (blockIdx.x >> 3);             // XCD k walks the k-th eighth of the tile list
  if ((blockIdx.x >> 3) >= per || sup >= nsuper) return;
  const unsigned lin0 = sup * C1_TPW;
  const int ntile = (int)((total - lin0) < C1_TPW ? (total - lin0) : C1_TPW);
  // columns 20 - s .. 19 of copy s are never written (no halo column behind them; zero-weight K slots read them): zero once
  for (int i = tid; i < 2 * C1H_IMG_ALLOC / 8; i += 256) reinterpret_cast<uint4*>(&halo[0][0])[i] = make_uint4(0, 0, 0, 0);
  C1Conv cv;
  cv.setup(w, l16, g, wv);
  C1Halo hl;
  hl.setup(tid, H, W);
  float hreg[5];
  unsigned ftx = lin0 % nx, fty = (lin0 / nx) % ny, fbt = lin0 / (nx * ny), ft = fbt % (unsigned)T;      // the tile being fetched
  auto fetch = [&]() __attribute__((always_inline)) {
    hl.fetch(x, tid, (int)fbt, (int)ft, (int)fty * 16, (int)ftx * 16, T, H, W, hreg);
    if (++ftx == nx) { ftx = 0; if (++fty == ny) { fty = 0; ++fbt; if (++ft == (unsigned)T) ft = 0; } }
  };
  auto stash = [&](int buf) __attribute__((always_inline)) {
    unsigned short h16[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) h16[j] = Mma<MODE_F16>::cvt(hreg[j]);
    hl.stash(&halo[buf][0], tid, h16);
  };
  fetch();
  __syncthreads();                 // the zero fill is complete
  stash(0);
  __syncthreads();
  unsigned ctx = lin0 % nx, cty = (lin0 / nx) % ny, cbt = lin0 / (nx * ny);    // the tile being computed
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  bool store = EPI == 0;
  float sc[4] = {0.f, 0.f, 0.f, 0.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (EPI == 1) {
    if (y != nullptr)
      for (int c = 0; c < 16; ++c) store |= fabsf(ep.gamma[c]) < C1_MIN_GAMMA;      // uniform over the grid
  }
  if constexpr (EPI == 2) {
    // the expressions of bn_pool_act_fwd_kernel, channels 4 g .. 4 g + 3
    const float4 ga = *reinterpret_cast<const float4*>(ep.gamma + 4 * g), is = *reinterpret_cast<const float4*>(ep.invstd + 4 * g);
    const float4 be = *reinterpret_cast<const float4*>(ep.beta + 4 * g), mu = *reinterpret_cast<const float4*>(ep.mean + 4 * g);
    sc[0] = ga.x * is.x; sc[1] = ga.y * is.y; sc[2] = ga.z * is.z; sc[3] = ga.w * is.w;
    sh[0] = be.x - mu.x * sc[0]; sh[1] = be.y - mu.y * sc[1]; sh[2] = be.z - mu.z * sc[2]; sh[3] = be.w - mu.w * sc[3];
  }
  for (int kt = 0; kt < ntile; ++kt) {
    const int bt = (int)cbt, x0 = (int)ctx * 16, y0 = (int)cty * 16;
    if (++ctx == nx) { ctx = 0; if (++cty == ny) { cty = 0; ++cbt; } }
    if (kt + 1 < ntile) fetch();
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    cv.tile(&halo[kt & 1][0], acc);
    // ---- lane holds channels 4 g + (0..3) of position (row 4 wv + i, column l16)
    if constexpr (EPI != 2) {
      // 16-byte stores, the 16 lanes of a row group write 16 consecutive positions = 1 KiB contiguous per store instruction
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int oy = y0 + 4 * wv + i, ox = x0 + l16;
        if (oy < H && ox < W) {
          if (store) *reinterpret_cast<float4*>(y + (((int64_t)bt * H + oy) * W + ox) * 16 + 4 * g) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
#pragma unroll
          for (int r = 0; r < 4; ++r) { s1[r] += acc[i][r]; s2[r] += acc[i][r] * acc[i][r]; }
        }
      }
    } else {
      // 2x2 windows: rows (4 wv + 0, 1) and (4 wv + 2, 3) live in this lane, columns (l16 even, odd) in a lane pair.  The even
      // lane finishes the upper window, the odd lane the lower one: each sends the partner the two rows it does not finish.
      const bool odd = (l16 & 1) != 0;
      float own[2][4], rcv[2][4];
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float va = acc[k][r] * sc[r] + sh[r], vb = acc[2 + k][r] * sc[r] + sh[r];
          own[k][r] = odd ? vb : va;
          rcv[k][r] = dpp_f32<0xB1, 0xf>(odd ? va : vb, 0.f);                       // quad_perm [1,0,3,2]: the pair partner's value
        }
      const int py = (y0 >> 1) + 2 * wv + (odd ? 1 : 0), px = (x0 >> 1) + (l16 >> 1);
      if (py < ep.Hp && px < ep.Wp) {
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float v = (dx == 1) == odd ? own[dy][r] : rcv[dy][r];
              if (v > best[r] || (v != v && best[r] == best[r])) { best[r] = v; bi[r] = dy * 2 + dx; }
            }
        float a4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) a4[r] = best[r] > 0.f ? best[r] : 0.01f * best[r];
        const int64_t pp = (((int64_t)bt * ep.Hp + py) * ep.Wp + px) * 16 + 4 * g;
        *reinterpret_cast<float4*>(ep.out + pp) = make_float4(a4[0], a4[1], a4[2], a4[3]);
        if (ep.out16 != nullptr) *reinterpret_cast<uint2*>(ep.out16 + pp) = make_uint2(pack2<2>(a4[0], a4[1]), pack2<2>(a4[2], a4[3]));
        if (ep.out_bf16 != nullptr) *reinterpret_cast<uint2*>(ep.out_bf16 + pp) = make_uint2(pack2<0>(a4[0], a4[1]), pack2<0>(a4[2], a4[3]));
        *reinterpret_cast<uchar4*>(ep.argmax + pp) = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
      }
    }
    if (kt + 1 < ntile) stash((kt + 1) & 1);
    __syncthreads();
  }
  if (EPI != 2 && stat_partials != nullptr) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s1[r] = row16_sum(s1[r]);
      s2[r] = row16_sum(s2[r]);
      if (l16 == 0) { red[wv][0][4 * g + r] = s1[r]; red[wv][1][4 * g + r] = s2[r]; }
    }
    __syncthreads();
    if (tid < 32) {
      const int which = tid >> 4, c = tid & 15;
      stat_partials[(int64_t)sup * 32 + tid] = red[0][which][c] + red[1][which][c] + red[2][which][c] + red[3][which][c];
    }
  }
}

// reference layout [16][1][3][5][5] -> [75][16]
__global__ void conv3d_c1_prep_kernel(const float* __restrict__ w, float* __restrict__ w16) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 1200) w16[i] = w[(i % 16) * 75 + i / 16];
}

// dW[16][75] partial per block; dy tile and x halo in LDS.  Thread = ((kd,kh) pair, 4-channel group, row worker): it walks
// rows of the 16x16 tile keeping the five x values of the kw window in registers (one new LDS value per position) and
// reading its 4 dy channels once per position -- 20 FMAs per 2 LDS reads.  (The first version, one thread per tap
// reading all 16 channels, spent 5 LDS reads per 16 FMAs and was LDS-bound at 17 us per tile.)
// FUSE_BN: `dy` is the pre-BatchNorm conv output y and the gradient is formed on the way into LDS from the pooled
// gradient / output / argmax and the BatchNorm backward coefficients -- the bn_pool_act_bwd_dx pass of the first layer
// (whose only consumer is this kernel: the network input needs no gradient) and its 1.6 GB dy round trip disappear.
// x / pool for pool = 2 or 3 (x >= 0, x < 98304) without the runtime integer division (~25 vector instructions each, sixteen of them
// per wave and tile in the first layer's weight-gradient kernels: 768 -> 697 us for the recompute kernel)
__device__ __forceinline__ int c1_pdiv(int x, int pool) { return pool == 2 ? (x >> 1) : (int)(((unsigned)x * 43691u) >> 17); }

struct C1BnArgs {
  const float* dout;            // [BT][Hp][Wp][16] gradient of the pooled, activated output
  const float* out;             // [BT][Hp][Wp][16] that output
  const unsigned char* argmax;  // [BT][Hp][Wp][16] window position of the maximum
  const float* mean;
  const float* invstd;
  const float* coef;            // [3][16]: gamma*invstd, mean(dz), mean(dz*xhat)   (bn_bwd_finalize_kernel)
  const float* beta;            // recompute kernel only: the sign of the pooled output is re-derived from the recomputed y
  int pool, Hp, Wp;
};
template <bool FUSE_BN>
__global__ __launch_bounds__(256) void conv3d_c1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ partials, int T, int H, int W,
                                                              int tiles_x, int tiles_y, int BT, int tiles_per_chunk, int nchunk,
                                                              C1BnArgs bn) {
  __shared__ float halo[3][20][21];
  __shared__ __attribute__((aligned(16))) float buf[4 * 1200];   // dy tile [256][16]; at the end the cross-worker reduction [4][1200]
  float (*dys)[16] = reinterpret_cast<float (*)[16]>(buf);
  const int tid = threadIdx.x;
  const bool active = tid < 240;
  const int worker = tid / 60, q = tid % 60, khd = q >> 2, c4 = (q & 3) * 4;
  const int kd = khd / 5, kh = khd % 5;
  float acc[5][4];
#pragma unroll
  for (int k = 0; k < 5; ++k)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[k][c] = 0.f;
  const int tiles_total = BT * tiles_x * tiles_y;
  const int chunk = (blockIdx.x & 7) * ((nchunk + 7) / 8) + (blockIdx.x >> 3);   // contiguous chunk range per XCD (frame re-reads meet in its L2)
  if (chunk >= nchunk) return;
  const int tile_beg = chunk * tiles_per_chunk, tile_end = min(tiles_total, tile_beg + tiles_per_chunk);
  for (int tile = tile_beg; tile < tile_end; ++tile) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, bt = tile / (tiles_x * tiles_y), t = bt % T;
    const int x0 = tx * 16, y0 = ty * 16;
    __syncthreads();
    for (int i = tid; i < 1200; i += 256) {
      const int d = i / 400, r = (i % 400) / 20, c = i % 20;
      const int tt = t + d - 1, iy = y0 + r - 2, ix = x0 + c - 2;
      float v = 0.f;
      if (tt >= 0 && tt < T && iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[((int64_t)(bt + d - 1) * H + iy) * W + ix];
      halo[d][r][c] = v;
    }
    for (int i = tid; i < 1024; i += 256) {
      const int pos = i >> 2, cc = (i & 3) * 4;
      const int oy = y0 + (pos >> 4), ox = x0 + (pos & 15);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (oy < H && ox < W) {
        v = *reinterpret_cast<const float4*>(dy + (((int64_t)bt * H + oy) * W + ox) * 16 + cc);
        if constexpr (FUSE_BN) {
          // same arithmetic as bn_pool_act_bwd_dx_kernel (bn_pool.hip), LeakyReLU(0.01) + max pool
          const int py = c1_pdiv(oy, bn.pool), px = c1_pdiv(ox, bn.pool);
          float gg[4] = {0.f, 0.f, 0.f, 0.f};
          if (py < bn.Hp && px < bn.Wp) {
            const int64_t pp = (((int64_t)bt * bn.Hp + py) * bn.Wp + px) * 16 + cc;
            const int here = (oy - py * bn.pool) * bn.pool + (ox - px * bn.pool);
            const uchar4 am = *reinterpret_cast<const uchar4*>(bn.argmax + pp);
            if (am.x == here || am.y == here || am.z == here || am.w == here) {
              const float4 dv = *reinterpret_cast<const float4*>(bn.dout + pp), ov = *reinterpret_cast<const float4*>(bn.out + pp);
              if (am.x == here) gg[0] = dv.x * (ov.x > 0.f ? 1.f : 0.01f);
              if (am.y == here) gg[1] = dv.y * (ov.y > 0.f ? 1.f : 0.01f);
              if (am.z == here) gg[2] = dv.z * (ov.z > 0.f ? 1.f : 0.01f);
              if (am.w == here) gg[3] = dv.w * (ov.w > 0.f ? 1.f : 0.01f);
            }
          }
          const float4 mu = *reinterpret_cast<const float4*>(bn.mean + cc), is = *reinterpret_cast<const float4*>(bn.invstd + cc);
          const float4 k0 = *reinterpret_cast<const float4*>(bn.coef + cc), k1 = *reinterpret_cast<const float4*>(bn.coef + 16 + cc);
          const float4 k2 = *reinterpret_cast<const float4*>(bn.coef + 32 + cc);
          v.x = k0.x * (gg[0] - k1.x - (v.x - mu.x) * is.x * k2.x);
          v.y = k0.y * (gg[1] - k1.y - (v.y - mu.y) * is.y * k2.y);
          v.z = k0.z * (gg[2] - k1.z - (v.z - mu.z) * is.z * k2.z);
          v.w = k0.w * (gg[3] - k1.w - (v.w - mu.w) * is.w * k2.w);
        }
      }
      *reinterpret_cast<float4*>(&dys[pos][cc]) = v;
    }
    __syncthreads();
    if (active) {
      for (int ry = worker; ry < 16; ry += 4) {
        const float* xr = &halo[kd][ry + kh][0];
        float w[5] = {xr[0], xr[1], xr[2], xr[3], 0.f};
#pragma unroll
        for (int p = 0; p < 16; ++p) {
          w[4] = xr[p + 4];
          const float4 d = *reinterpret_cast<const float4*>(&dys[ry * 16 + p][c4]);
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            acc[k][0] = fmaf(w[k], d.x, acc[k][0]);
            acc[k][1] = fmaf(w[k], d.y, acc[k][1]);
            acc[k][2] = fmaf(w[k], d.z, acc[k][2]);
            acc[k][3] = fmaf(w[k], d.w, acc[k][3]);
          }
          w[0] = w[1]; w[1] = w[2]; w[2] = w[3]; w[3] = w[4];
        }
      }
    }
  }
  __syncthreads();
  float* red = buf;                        // [4 workers][16 c][75 taps]
  if (active)
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
      for (int c = 0; c < 4; ++c) red[(worker * 16 + c4 + c) * 75 + khd * 5 + k] = acc[k][c];
  __syncthreads();
  for (int i = tid; i < 1200; i += 256)     // i = c * 75 + tap: [chunk][c][tap]
    partials[(int64_t)chunk * 1200 + i] = red[i] + red[1200 + i] + red[2400 + i] + red[3600 + i];
}

// The first layer's weight gradient on the matrix pipe (16-bit path), BatchNorm / max-pool / LeakyReLU backward fused as in
// conv3d_c1_wgrad_kernel<true>:  dW[tap][co] = sum over positions of x[pos + tap] * dy[pos][co]  as an MFMA product with the
// 75 taps as rows (5 tiles of 16, the last 5 rows unused), co as columns and the POSITION as the K dimension (32 positions =
// two tile rows per step).  Per 16x16 tile: the x halo is staged as bf16 [3][20][24]; dy is formed in f32 from the pooled
// gradient exactly as before, rounded to bf16 and stored TRANSPOSED [co][pos] so that a B fragment (8 consecutive positions
// of one channel) is one 16-byte LDS read; an A fragment of lane (tap, k group) is 8 consecutive halo columns of the tap's
// (kd, kh) row starting at column kw (eight 16-bit reads).  A wave takes 2 of the tile's 8 K-steps for all 5 tap tiles and
// keeps its 5 accumulators across the chunk's tiles; the waves are summed once per chunk.  75 FMAs per (position, channel)
// on the VALU become 5 MFMAs per 32 positions: the f32 kernel took 1.55 ms per step, this one is bound by reading y (1.6 GB).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv3d_c1_wgrad_mfma_kernel(const float* __restrict__ x, const float* __restrict__ yconv,
                                                                   float* __restrict__ partials, int T, int H, int W, int tiles_x,
                                                                   int tiles_y, int BT, int tiles_per_chunk, int nchunk, C1BnArgs bn) {
  constexpr int DYS = 256 + 8;                                        // dyT row stride (elements): 528 B, 16-byte aligned
  __shared__ __attribute__((aligned(16))) char smem[4 * 80 * 16 * 4];  // max(halo + dyT = 2880 + 8448 B, final reduction 20480 B)
  unsigned short (*halo)[20][24] = reinterpret_cast<unsigned short (*)[20][24]>(smem);
  unsigned short* dyT = reinterpret_cast<unsigned short*>(smem + 3 * 20 * 24 * 2);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l16 = lane & 15, g = lane >> 4;
  const int tiles_total = BT * tiles_x * tiles_y;
  const int chunk = (blockIdx.x & 7) * ((nchunk + 7) / 8) + (blockIdx.x >> 3);
  if (chunk >= nchunk) return;
  // per-lane halo byte offsets of the 5 tap tiles: tap (kd, kh, kw), plus this lane's k group (row g >> 1, column 8 (g & 1))
  int abase[5];
#pragma unroll
  for (int mt = 0; mt < 5; ++mt) {
    int tap = 16 * mt + l16;
    tap = tap < 75 ? tap : 74;                                       // rows 75..79 of the last tile: computed, never stored
    const int kd = tap / 25, kh = (tap % 25) / 5, kw = tap % 5;
    abase[mt] = (((kd * 20 + kh + (g >> 1)) * 24) + kw + 8 * (g & 1)) * 2;
  }
  f32x4 acc[5];
#pragma unroll
  for (int mt = 0; mt < 5; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int tile_beg = chunk * tiles_per_chunk, tile_end = min(tiles_total, tile_beg + tiles_per_chunk);
  for (int tile = tile_beg; tile < tile_end; ++tile) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, bt = tile / (tiles_x * tiles_y), t = bt % T;
    const int x0 = tx * 16, y0 = ty * 16;
    __syncthreads();
    // dy formation: all of a thread's 16 loads (4 positions x {y, argmax, dout, out}) are issued before the first is used, ahead of the halo's loads --
    // with the conditional, dependent loads of the f32 kernel's loop a tile took 12 us of chained memory round trips.
    // Coordinates are clamped instead of predicated (edge tiles: the values are discarded below).
    float4 yv[4], dv[4], ov[4];
    uchar4 am[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = tid + it * 256, pos = i >> 2, cc = (i & 3) * 4;
      const int oy = min(y0 + (pos >> 4), H - 1), ox = min(x0 + (pos & 15), W - 1);
      const int py = min(c1_pdiv(oy, bn.pool), bn.Hp - 1), px = min(c1_pdiv(ox, bn.pool), bn.Wp - 1);
      const int64_t pp = (((int64_t)bt * bn.Hp + py) * bn.Wp + px) * 16 + cc;
      yv[it] = *reinterpret_cast<const float4*>(yconv + (((int64_t)bt * H + oy) * W + ox) * 16 + cc);
      am[it] = *reinterpret_cast<const uchar4*>(bn.argmax + pp);
      dv[it] = *reinterpret_cast<const float4*>(bn.dout + pp);
      ov[it] = *reinterpret_cast<const float4*>(bn.out + pp);
    }
    for (int i = tid; i < 1200; i += 256) {
      const int d = i / 400, r = (i % 400) / 20, c = i % 20;
      const int tt = t + d - 1, iy = y0 + r - 2, ix = x0 + c - 2;
      float v = 0.f;
      if (tt >= 0 && tt < T && iy >= 0 && iy < H && ix >= 0 && ix < W) v = x[((int64_t)(bt + d - 1) * H + iy) * W + ix];
      halo[d][r][c] = f2bf(v);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = tid + it * 256, pos = i >> 2, cc = (i & 3) * 4;
      const int oy = y0 + (pos >> 4), ox = x0 + (pos & 15);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (oy < H && ox < W) {
        // same arithmetic as bn_pool_act_bwd_dx_kernel (bn_pool.hip), LeakyReLU(0.01) + max pool
        const int py = c1_pdiv(oy, bn.pool), px = c1_pdiv(ox, bn.pool);
        float gg[4] = {0.f, 0.f, 0.f, 0.f};
        if (py < bn.Hp && px < bn.Wp) {
          const int here = (oy - py * bn.pool) * bn.pool + (ox - px * bn.pool);
          if (am[it].x == here) gg[0] = dv[it].x * (ov[it].x > 0.f ? 1.f : 0.01f);
          if (am[it].y == here) gg[1] = dv[it].y * (ov[it].y > 0.f ? 1.f : 0.01f);
          if (am[it].z == here) gg[2] = dv[it].z * (ov[it].z > 0.f ? 1.f : 0.01f);
          if (am[it].w == here) gg[3] = dv[it].w * (ov[it].w > 0.f ? 1.f : 0.01f);
        }
        const float4 mu = *reinterpret_cast<const float4*>(bn.mean + cc), is = *reinterpret_cast<const float4*>(bn.invstd + cc);
        const float4 k0 = *reinterpret_cast<const float4*>(bn.coef + cc), k1 = *reinterpret_cast<const float4*>(bn.coef + 16 + cc);
        const float4 k2 = *reinterpret_cast<const float4*>(bn.coef + 32 + cc);
        v.x = k0.x * (gg[0] - k1.x - (yv[it].x - mu.x) * is.x * k2.x);
        v.y = k0.y * (gg[1] - k1.y - (yv[it].y - mu.y) * is.y * k2.y);
        v.z = k0.z * (gg[2] - k1.z - (yv[it].z - mu.z) * is.z * k2.z);
        v.w = k0.w * (gg[3] - k1.w - (yv[it].w - mu.w) * is.w * k2.w);
      }
      dyT[(cc + 0) * DYS + pos] = f2bf(v.x);
      dyT[(cc + 1) * DYS + pos] = f2bf(v.y);
      dyT[(cc + 2) * DYS + pos] = f2bf(v.z);
      dyT[(cc + 3) * DYS + pos] = f2bf(v.w);
    }
    __syncthreads();
    const char* hb0 = reinterpret_cast<const char*>(&halo[0][0][0]);
#pragma unroll
    for (int ksl = 0; ksl < 2; ++ksl) {
      const int ks = wv * 2 + ksl;                                   // positions 32 ks .. 32 ks + 31 = tile rows 2 ks, 2 ks + 1
      const bf16x8 fb = *reinterpret_cast<const bf16x8*>(dyT + l16 * DYS + 32 * ks + 8 * g);
      const char* hb = hb0 + ks * (2 * 24 * 2);
#pragma unroll
      for (int mt = 0; mt < 5; ++mt) {
        const unsigned short* ap = reinterpret_cast<const unsigned short*>(hb + abase[mt]);
        const unsigned a0 = ap[0] | ((unsigned)ap[1] << 16), a1 = ap[2] | ((unsigned)ap[3] << 16);
        const unsigned a2 = ap[4] | ((unsigned)ap[5] << 16), a3 = ap[6] | ((unsigned)ap[7] << 16);
        Mma<MODE_BF16>::mma(acc[mt], __builtin_bit_cast(bf16x8, make_uint4(a0, a1, a2, a3)), fb);
      }
    }
  }
  // ---- sum the four waves: lane (co = l16, g) holds taps 16 mt + 4 g + r
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);                        // [4 waves][80 taps][16 co]
#pragma unroll
  for (int mt = 0; mt < 5; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wv * 80 + 16 * mt + 4 * g + r) * 16 + l16] = acc[mt][r];
  __syncthreads();
  for (int i = tid; i < 1200; i += 256) {     // i = c * 75 + tap: [chunk][c][tap]
    const int c = i / 75, tap = i % 75;
    partials[(int64_t)chunk * 1200 + i] = red[tap * 16 + c] + red[(80 + tap) * 16 + c] + red[(160 + tap) * 16 + c] + red[(240 + tap) * 16 + c];
  }
}

// conv3d_c1_wgrad_recompute_kernel: the same product WITHOUT the stored conv output: the tile's y is recomputed from the halo that is
// staged anyway (C1Conv on an IEEE-half image of it: the arithmetic of conv3d_c1_fwd_mfma_kernel, so xhat is bit-identical to what the
// forward normalised).  The MFMA result layout -- lane (column l16, channel group g) holds channels 4 g .. 4 g + 3 of the positions
// (row 4 wv + i, l16) -- is also the dy-formation mapping: each lane forms its 4 x 4 values from registers and writes them transposed.
// The bf16 halo of the weight-gradient product is kept in the same four-shifted-copies form: the A fragment of lane (tap, k group) is 8
// consecutive columns starting at kw + 8 (g & 1), i.e. copy kw & 3 at an 8-byte aligned column -- one ds_read2_b64 instead of eight
// 16-bit reads and four packs.  Both images are double-buffered and the next tile's halo is fetched a tile ahead.
__global__ __launch_bounds__(256) void conv3d_c1_wgrad_recompute_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                        float* __restrict__ partials, int T, int H, int W, int tiles_x,
                                                                        int tiles_y, int BT, int tiles_per_chunk, int nchunk, C1BnArgs bn) {
  constexpr int DYS = 256 + 8;
  // [2 buffers][half image | bf16 image] + dyT; the final reduction (20480 B) reuses the front
  __shared__ __attribute__((aligned(16))) unsigned short img[2][2][C1H_IMG_ALLOC];
  __shared__ __attribute__((aligned(16))) unsigned short dyT[16 * DYS];
  static_assert(sizeof(unsigned short) * 4 * C1H_IMG_ALLOC >= 4 * 80 * 16 * 4, "reduction scratch");
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l16 = lane & 15, g = lane >> 4;
  const int tiles_total = BT * tiles_x * tiles_y;
  const int chunk = (blockIdx.x & 7) * ((nchunk + 7) / 8) + (blockIdx.x >> 3);
  if (chunk >= nchunk) return;
  for (int i = tid; i < 4 * C1H_IMG_ALLOC / 8; i += 256) reinterpret_cast<uint4*>(&img[0][0][0])[i] = make_uint4(0, 0, 0, 0);
  // byte offsets of the A fragments of the 5 tap tiles (K step ks: add 80 ks)
  unsigned abase[5];
#pragma unroll
  for (int mt = 0; mt < 5; ++mt) {
    int tap = 16 * mt + l16;
    tap = tap < 75 ? tap : 74;                                       // rows 75..79 of the last tile: computed, never stored
    const int kd = tap / 25, kh = (tap % 25) / 5, kw = tap % 5;
    abase[mt] = (unsigned)(((kw & 3) * C1H_COPY + (kd * 20 + kh + (g >> 1)) * 20 + (kw & ~3) + 8 * (g & 1)) * 2);
  }
  C1Conv cv;
  cv.setup(w, l16, g, wv);
  C1Halo hl;
  hl.setup(tid, H, W);
  // the per-channel constants of the dy formula live in LDS (24 registers otherwise, which cost the third wave per SIMD)
  __shared__ __attribute__((aligned(16))) float cst[6][16];          // mean, invstd, gamma invstd, mean(dz), mean(dz xhat), beta - mean gamma invstd
  if (tid < 16) {
    const float m_ = bn.mean[tid], c0_ = bn.coef[tid];
    cst[0][tid] = m_; cst[1][tid] = bn.invstd[tid]; cst[2][tid] = c0_; cst[3][tid] = bn.coef[16 + tid]; cst[4][tid] = bn.coef[32 + tid];
    cst[5][tid] = bn.beta[tid] - m_ * c0_;
  }
  f32x4 acc[5];
#pragma unroll
  for (int mt = 0; mt < 5; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int tile_beg = chunk * tiles_per_chunk, tile_end = min(tiles_total, tile_beg + tiles_per_chunk);
  if (tile_beg >= tile_end) {                                        // an empty chunk still owns its row of partials
    for (int i = tid; i < 1200; i += 256) partials[(int64_t)chunk * 1200 + i] = 0.f;
    return;
  }
  int tx = tile_beg % tiles_x, ty = (tile_beg / tiles_x) % tiles_y, bt = tile_beg / (tiles_x * tiles_y), t = bt % T;
  float hreg[5];
  auto stash = [&](int buf) __attribute__((always_inline)) {
    unsigned short h16[5], b16[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) { h16[j] = Mma<MODE_F16>::cvt(hreg[j]); b16[j] = f2bf(hreg[j]); }
    hl.stash(&img[buf][0][0], tid, h16);
    hl.stash(&img[buf][1][0], tid, b16);
  };
  hl.fetch(x, tid, bt, t, ty * 16, tx * 16, T, H, W, hreg);
  __syncthreads();                                                   // zero fill complete
  stash(0);
  __syncthreads();
  for (int tile = tile_beg, it = 0; tile < tile_end; ++tile, ++it) {
    const int x0 = tx * 16, y0 = ty * 16, cbt = bt;
    if (++tx == tiles_x) { tx = 0; if (++ty == tiles_y) { ty = 0; ++bt; if (++t == T) t = 0; } }
    // the pooled operands of this lane's four positions and the next tile's halo: issued first, used after the conv
    float4 dv[4];
    uchar4 am[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int oy = min(y0 + 4 * wv + i, H - 1), ox = min(x0 + l16, W - 1);
      const int py = min(c1_pdiv(oy, bn.pool), bn.Hp - 1), px = min(c1_pdiv(ox, bn.pool), bn.Wp - 1);
      const int64_t pp = (((int64_t)cbt * bn.Hp + py) * bn.Wp + px) * 16 + 4 * g;
      am[i] = *reinterpret_cast<const uchar4*>(bn.argmax + pp);
      dv[i] = *reinterpret_cast<const float4*>(bn.dout + pp);
    }
    if (tile + 1 < tile_end) hl.fetch(x, tid, bt, t, ty * 16, tx * 16, T, H, W, hreg);
    // ---- y of the tile (forward arithmetic)
    f32x4 z[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    cv.tile(&img[it & 1][0][0], z);
    // ---- dy of this lane's positions (the arithmetic of bn_pool_act_bwd_dx_kernel), transposed bf16 into LDS
    const float4 mu = *reinterpret_cast<const float4*>(&cst[0][4 * g]), is = *reinterpret_cast<const float4*>(&cst[1][4 * g]);
    const float4 k0 = *reinterpret_cast<const float4*>(&cst[2][4 * g]), k1 = *reinterpret_cast<const float4*>(&cst[3][4 * g]);
    const float4 k2 = *reinterpret_cast<const float4*>(&cst[4][4 * g]), sh = *reinterpret_cast<const float4*>(&cst[5][4 * g]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int oy = y0 + 4 * wv + i, ox = x0 + l16, pos = (4 * wv + i) * 16 + l16;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (oy < H && ox < W) {
        const int py = c1_pdiv(oy, bn.pool), px = c1_pdiv(ox, bn.pool);
        float gg[4] = {0.f, 0.f, 0.f, 0.f};
        if (py < bn.Hp && px < bn.Wp) {
          const int here = (oy - py * bn.pool) * bn.pool + (ox - px * bn.pool);
          // the pooled output is positive exactly when the forward's pre-activation at the argmax was: y * (gamma invstd) + (beta - mean gamma invstd),
          // the expression of conv3d_c1_fwd_mfma_kernel<2> on the same recomputed y -- `out` is not read
          if (am[i].x == here) gg[0] = dv[i].x * (z[i][0] * k0.x + sh.x > 0.f ? 1.f : 0.01f);
          if (am[i].y == here) gg[1] = dv[i].y * (z[i][1] * k0.y + sh.y > 0.f ? 1.f : 0.01f);
          if (am[i].z == here) gg[2] = dv[i].z * (z[i][2] * k0.z + sh.z > 0.f ? 1.f : 0.01f);
          if (am[i].w == here) gg[3] = dv[i].w * (z[i][3] * k0.w + sh.w > 0.f ? 1.f : 0.01f);
        }
        v.x = k0.x * (gg[0] - k1.x - (z[i][0] - mu.x) * is.x * k2.x);
        v.y = k0.y * (gg[1] - k1.y - (z[i][1] - mu.y) * is.y * k2.y);
        v.z = k0.z * (gg[2] - k1.z - (z[i][2] - mu.z) * is.z * k2.z);
        v.w = k0.w * (gg[3] - k1.w - (z[i][3] - mu.w) * is.w * k2.w);
      }
      dyT[(4 * g + 0) * DYS + pos] = f2bf(v.x);
      dyT[(4 * g + 1) * DYS + pos] = f2bf(v.y);
      dyT[(4 * g + 2) * DYS + pos] = f2bf(v.z);
      dyT[(4 * g + 3) * DYS + pos] = f2bf(v.w);
    }
    __syncthreads();
    const char* hb0 = reinterpret_cast<const char*>(&img[it & 1][1][0]);
#pragma unroll
    for (int ksl = 0; ksl < 2; ++ksl) {
      const int ks = wv * 2 + ksl;                                   // positions 32 ks .. 32 ks + 31 = tile rows 2 ks, 2 ks + 1
      const bf16x8 fb = *reinterpret_cast<const bf16x8*>(dyT + l16 * DYS + 32 * ks + 8 * g);
      const char* hb = hb0 + ks * 80;
#pragma unroll
      for (int mt = 0; mt < 5; ++mt) {
        const uint2 lo = *reinterpret_cast<const uint2*>(hb + abase[mt]), hi = *reinterpret_cast<const uint2*>(hb + abase[mt] + 8);
        Mma<MODE_BF16>::mma(acc[mt], __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y)), fb);
      }
    }
    if (tile + 1 < tile_end) stash((it + 1) & 1);
    __syncthreads();
  }
  // ---- sum the four waves: lane (co = l16, g) holds taps 16 mt + 4 g + r
  float* red = reinterpret_cast<float*>(&img[0][0][0]);              // [4 waves][80 taps][16 co]
#pragma unroll
  for (int mt = 0; mt < 5; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wv * 80 + 16 * mt + 4 * g + r) * 16 + l16] = acc[mt][r];
  __syncthreads();
  for (int i = tid; i < 1200; i += 256) {     // i = c * 75 + tap: [chunk][c][tap]
    const int c = i / 75, tap = i % 75;
    partials[(int64_t)chunk * 1200 + i] = red[tap * 16 + c] + red[(80 + tap) * 16 + c] + red[(160 + tap) * 16 + c] + red[(240 + tap) * 16 + c];
  }
}

__global__ __launch_bounds__(256) void conv3d_c1_wgrad_reduce_kernel(const float* __restrict__ partials, float* __restrict__ dw, int nchunk, int beta) {
  bool owner;
  int i;
  const float s = chunk_sum16(partials, 1200, nchunk, blockIdx.x * 16, owner, i);
  if (owner) dw[i] = beta ? dw[i] + s : s;
}

// rows of `stat_partials` ([rows][2][16] floats) maavss_conv3d_c1_fwd writes: one per workgroup
extern "C" int64_t maavss_conv3d_c1_fwd_nparts(int B, int T, int H, int W, int precise) {
  const int64_t tiles = (int64_t)cdiv(W, 16) * cdiv(H, 16) * B * T;
  return precise == MODE_F16 ? cdiv(tiles, C1_TPW) : tiles;
}

extern "C" int maavss_conv3d_c1_fwd(const float* x, const float* w, float* w16_ws, float* y, float* stat_partials, int B,
                                    int T, int H, int W, int precise, void* stream) {
  MAAVSS_CHECK_ARG(x && w && w16_ws && y, "conv3d_c1_fwd: null pointer");
  MAAVSS_CHECK_ARG(B > 0 && T > 0 && H > 0 && W > 0, "conv3d_c1_fwd: empty problem");
  MAAVSS_CHECK_ARG((int64_t)cdiv(W, 16) * cdiv(H, 16) * B * T < (1LL << 31) && (int64_t)B * T * H * W < (1LL << 40), "conv3d_c1: too many tiles");
  MAAVSS_CHECK_ARG(precise == MODE_F32 || precise == MODE_F16, "conv3d_c1_fwd: mode must be 1 (exact f32 VALU) or 2 (IEEE-half MFMA)");
  hipStream_t st = (hipStream_t)stream;
  if (precise == MODE_F16) {
    hipLaunchKernelGGL(conv3d_c1_fwd_mfma_kernel<0>, dim3(xcd_grid(cdiv((int64_t)cdiv(W, 16) * cdiv(H, 16) * B * T, C1_TPW))), dim3(256), 0, st, x, w, y,
                       stat_partials, B * T, T, H, W, C1EpiArgs{});
    MAAVSS_LAUNCH_CHECK("conv3d_c1_fwd_mfma_kernel");
    return MAAVSS_OK;
  }
  hipLaunchKernelGGL(conv3d_c1_prep_kernel, dim3(5), dim3(256), 0, st, w, w16_ws);
  hipLaunchKernelGGL(conv3d_c1_fwd_kernel, dim3(xcd_grid((int64_t)cdiv(W, 16) * cdiv(H, 16) * B * T)), dim3(256), 0, st, x, w16_ws, y,
                     stat_partials, B * T, T, H, W);
  MAAVSS_LAUNCH_CHECK("conv3d_c1_fwd_kernel");
  return MAAVSS_OK;
}

// The 16-bit first layer without its conv output (see conv3d_c1_fwd_mfma_kernel): pass 1, BatchNorm partial sums.  `y` is written
// only when some |gamma[c]| < 1e-2 (the backward reduction then gathers xhat from it); stat_partials as maavss_conv3d_c1_fwd(.., 2).
extern "C" int maavss_conv3d_c1_stats(const float* x, const float* w, const float* gamma, float* y, float* stat_partials, int B, int T,
                                      int H, int W, void* stream) {
  MAAVSS_CHECK_ARG(x && w && gamma && y && stat_partials, "conv3d_c1_stats: null pointer");
  MAAVSS_CHECK_ARG(B > 0 && T > 0 && H > 0 && W > 0, "conv3d_c1_stats: empty problem");
  MAAVSS_CHECK_ARG((int64_t)cdiv(W, 16) * cdiv(H, 16) * B * T < (1LL << 31) && (int64_t)B * T * H * W < (1LL << 40), "conv3d_c1: too many tiles");
  C1EpiArgs ep = {};
  ep.gamma = gamma;
  hipLaunchKernelGGL(conv3d_c1_fwd_mfma_kernel<1>, dim3(xcd_grid(cdiv((int64_t)cdiv(W, 16) * cdiv(H, 16) * B * T, C1_TPW))), dim3(256), 0,
                     (hipStream_t)stream, x, w, y, stat_partials, B * T, T, H, W, ep);
  MAAVSS_LAUNCH_CHECK("conv3d_c1_fwd_mfma_kernel<1>");
  return MAAVSS_OK;
}

// pass 2: conv again -> BatchNorm -> MaxPool(1,2,2) -> LeakyReLU(0.01).  out [B*T][H/2][W/2][16] f32, out16 the same as IEEE half
// and out_bf16 as bf16 (both may be null), argmax one byte per element.
extern "C" int maavss_conv3d_c1_bn_pool_act(const float* x, const float* w, const float* mean, const float* invstd, const float* gamma,
                                            const float* beta, float* out, void* out16, void* out_bf16, void* argmax, int B, int T, int H,
                                            int W, void* stream) {
  MAAVSS_CHECK_ARG(x && w && mean && invstd && gamma && beta && out && argmax, "conv3d_c1_bn_pool_act: null pointer");
  MAAVSS_CHECK_ARG(B > 0 && T > 0 && H >= 2 && W >= 2, "conv3d_c1_bn_pool_act: empty problem");
  MAAVSS_CHECK_ARG((int64_t)cdiv(W, 16) * cdiv(H, 16) * B * T < (1LL << 31) && (int64_t)B * T * H * W < (1LL << 40), "conv3d_c1: too many tiles");
  C1EpiArgs ep;
  ep.mean = mean; ep.invstd = invstd; ep.gamma = gamma; ep.beta = beta; ep.out = out; ep.out16 = (unsigned short*)out16;
  ep.out_bf16 = (unsigned short*)out_bf16;
  ep.argmax = (unsigned char*)argmax; ep.Hp = H / 2; ep.Wp = W / 2;
  hipLaunchKernelGGL(conv3d_c1_fwd_mfma_kernel<2>, dim3(xcd_grid(cdiv((int64_t)cdiv(W, 16) * cdiv(H, 16) * B * T, C1_TPW))), dim3(256), 0,
                     (hipStream_t)stream, x, w, nullptr, nullptr, B * T, T, H, W, ep);
  MAAVSS_LAUNCH_CHECK("conv3d_c1_fwd_mfma_kernel<2>");
  return MAAVSS_OK;
}

static int c1_wgrad_launch(const float* x, const float* dy_or_y, float* dw, float* ws, int nchunk, int B, int T, int H, int W, int beta,
                           const C1BnArgs* bn, hipStream_t st, bool mfma = false) {
  const int tiles_x = cdiv(W, 16), tiles_y = cdiv(H, 16);
  const int tiles_total = B * T * tiles_x * tiles_y;
  C1BnArgs none = {};
  if (bn && mfma)
    hipLaunchKernelGGL(conv3d_c1_wgrad_mfma_kernel, dim3(cdiv(nchunk, 8) * 8), dim3(256), 0, st, x, dy_or_y, ws, T, H, W, tiles_x, tiles_y,
                       B * T, cdiv(tiles_total, nchunk), nchunk, *bn);
  else if (bn)
    hipLaunchKernelGGL(conv3d_c1_wgrad_kernel<true>, dim3(cdiv(nchunk, 8) * 8), dim3(256), 0, st, x, dy_or_y, ws, T, H, W, tiles_x, tiles_y,
                       B * T, cdiv(tiles_total, nchunk), nchunk, *bn);
  else
    hipLaunchKernelGGL(conv3d_c1_wgrad_kernel<false>, dim3(cdiv(nchunk, 8) * 8), dim3(256), 0, st, x, dy_or_y, ws, T, H, W, tiles_x, tiles_y,
                       B * T, cdiv(tiles_total, nchunk), nchunk, none);
  MAAVSS_LAUNCH_CHECK("conv3d_c1_wgrad_kernel");
  hipLaunchKernelGGL(conv3d_c1_wgrad_reduce_kernel, dim3(75), dim3(256), 0, st, ws, dw, nchunk, beta);
  MAAVSS_LAUNCH_CHECK("conv3d_c1_wgrad_reduce_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_conv3d_c1_wgrad(const float* x, const float* dy, float* dw, float* ws, int nchunk, int B, int T,
                                      int H, int W, int beta, void* stream) {
  MAAVSS_CHECK_ARG(x && dy && dw && ws, "conv3d_c1_wgrad: null pointer");
  MAAVSS_CHECK_ARG(nchunk >= 1 && B > 0 && T > 0, "conv3d_c1_wgrad: bad sizes");
  MAAVSS_CHECK_ARG(H > 0 && W > 0 && (int64_t)cdiv(W, 16) * cdiv(H, 16) * B * T < (1LL << 31), "conv3d_c1_wgrad: empty image or too many tiles");
  return c1_wgrad_launch(x, dy, dw, ws, nchunk, B, T, H, W, beta, nullptr, (hipStream_t)stream);
}

// maavss_conv3d_c1_wgrad_bn on the 16-bit path without the stored conv output: `w` = the layer's weights [16][1][3][5][5], y is
// recomputed per tile (conv3d_c1_wgrad_recompute_kernel).  bf16 backward operands, IEEE-half forward operands for the recompute.
extern "C" int maavss_conv3d_c1_wgrad_bn_recompute(const float* x, const float* w, const float* dout, const void* argmax,
                                                   const float* mean, const float* invstd, const float* bn_beta, const float* coef, int pool,
                                                   float* dw, float* ws, int nchunk, int B, int T, int H, int W, int beta, void* stream) {
  MAAVSS_CHECK_ARG(x && w && dout && argmax && mean && invstd && bn_beta && coef && dw && ws, "conv3d_c1_wgrad_bn_recompute: null pointer");
  MAAVSS_CHECK_ARG(nchunk >= 1 && B > 0 && T > 0 && pool >= 2 && pool <= 3, "conv3d_c1_wgrad_bn_recompute: bad sizes (pool must be 2 or 3)");
  // H / pool >= 1 (the kernel clamps pooled indices to Hp - 1), c1_pdiv's multiply-shift division by 3 holds for x < 98304, the
  // tile count is an int
  MAAVSS_CHECK_ARG(H >= pool && W >= pool && H < 98304 && W < 98304, "conv3d_c1_wgrad_bn_recompute: H, W must be in [pool, 98304) (got %d x %d)", H, W);
  MAAVSS_CHECK_ARG((int64_t)cdiv(W, 16) * cdiv(H, 16) * B * T < (1LL << 31), "conv3d_c1_wgrad_bn_recompute: too many tiles");
  C1BnArgs bn;
  bn.dout = dout; bn.out = nullptr; bn.argmax = (const unsigned char*)argmax; bn.mean = mean; bn.invstd = invstd; bn.coef = coef;
  bn.beta = bn_beta; bn.pool = pool; bn.Hp = H / pool; bn.Wp = W / pool;
  const int tiles_x = cdiv(W, 16), tiles_y = cdiv(H, 16);
  const int tiles_total = B * T * tiles_x * tiles_y;
  hipLaunchKernelGGL(conv3d_c1_wgrad_recompute_kernel, dim3(cdiv(nchunk, 8) * 8), dim3(256), 0, (hipStream_t)stream, x, w, ws, T, H, W,
                     tiles_x, tiles_y, B * T, cdiv(tiles_total, nchunk), nchunk, bn);
  MAAVSS_LAUNCH_CHECK("conv3d_c1_wgrad_recompute_kernel");
  hipLaunchKernelGGL(conv3d_c1_wgrad_reduce_kernel, dim3(75), dim3(256), 0, (hipStream_t)stream, ws, dw, nchunk, beta);
  MAAVSS_LAUNCH_CHECK("conv3d_c1_wgrad_reduce_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_conv3d_c1_wgrad_bn(const float* x, const float* y, const float* dout, const float* out, const void* argmax,
                                         const float* mean, const float* invstd, const float* coef, int pool, float* dw, float* ws,
                                         int nchunk, int B, int T, int H, int W, int beta, int precise, void* stream) {
  MAAVSS_CHECK_ARG(x && y && dout && out && argmax && mean && invstd && coef && dw && ws, "conv3d_c1_wgrad_bn: null pointer");
  MAAVSS_CHECK_ARG(precise == MODE_F32 || precise == MODE_BF16, "conv3d_c1_wgrad_bn: mode must be 1 (exact f32 VALU) or 0 (bf16 MFMA)");
  MAAVSS_CHECK_ARG(nchunk >= 1 && B > 0 && T > 0 && pool >= 2 && pool <= 3, "conv3d_c1_wgrad_bn: bad sizes (pool must be 2 or 3)");
  MAAVSS_CHECK_ARG(H >= pool && W >= pool && H < 98304 && W < 98304 && (int64_t)cdiv(W, 16) * cdiv(H, 16) * B * T < (1LL << 31),
                   "conv3d_c1_wgrad_bn: H, W must be in [pool, 98304) and the tile count below 2^31 (got %d x %d)", H, W);
  C1BnArgs bn;
  bn.dout = dout; bn.out = out; bn.argmax = (const unsigned char*)argmax; bn.mean = mean; bn.invstd = invstd; bn.coef = coef;
  bn.beta = nullptr; bn.pool = pool; bn.Hp = H / pool; bn.Wp = W / pool;
  return c1_wgrad_launch(x, y, dw, ws, nchunk, B, T, H, W, beta, &bn, (hipStream_t)stream, precise == MODE_BF16);
}
