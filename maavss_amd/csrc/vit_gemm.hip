// K1/K3: the dense layers of the DINO ViT-S/8 (patch projection, qkv, proj, MLP fc1/fc2) as ONE bf16 MFMA GEMM
// with fused epilogues.  Replaces the Linear/Conv2d calls inside dino's VisionTransformer that the reference
// reaches through video_attention.py:52 (get_last_selfattention); the module is external to the reference
// (empty submodule), see oracle/vit_ref_cpu.py for the restated architecture.
//
//   C[M,N] = epilogue( A[M,K] (bf16, row-major) . W[N,K]^T (bf16, torch Linear layout) )
//
// 128x128 output tile per 256-thread workgroup (2x2 waves, 64x64 per wave = 4x4 MFMA 16x16x32 tiles), BK = 32.
// These GEMMs have SHORT K (384 / 1536 / 192), so the cost is latency per tile, not steady-state issue:
//   * operand tiles go global -> LDS directly (global_load_lds, 16 B per lane) into a 5-slot ring; up to three
//     K-steps stay in flight across a raw s_barrier behind a counted s_waitcnt vmcnt (never 0 in the loop);
//   * the LDS fragment reads of K-step kt+1 are issued BEFORE the 16 MFMAs of K-step kt (two named fragment
//     sets, loop unrolled by two), so LDS latency hides under the matrix pipe instead of in front of it;
//   * 80 KiB of LDS per workgroup -> two workgroups per CU overlap each other's prologue / epilogue;
//   * the LDS image is lane-linear, so the bank swizzle (16-B chunk ^= (row>>2)&2, conflict-free for the
//     ds_read_b128 lane groups on 64-B rows) is applied to the per-lane SOURCE address and again on the read;
//   * blocks are remapped so that the N-tiles of one M-panel run on the same XCD (A panel from that L2).
// Epilogues: +bias (and 1/8 on the q third) -> bf16 | +bias, GELU -> bf16 | +bias +residual -> f32 (in place
// on the residual stream) | + periodic row table (conv bias / cls token + position embedding) -> f32.
// All outputs are staged through LDS and written as full rows (16 B per lane).
#include "mma.h"

#define EPI_BF16_BIAS 0
#define EPI_BF16_BIAS_GELU 1
#define EPI_F32_BIAS_RESID 2
#define EPI_F32_ROWTABLE 3

#define VG_BK 32
#define VG_STAGES 5
#define VG_STAGE_ELEMS (2 * 128 * VG_BK)  // A tile + B tile, bf16 elements

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

template <int EPI>
__global__ __launch_bounds__(256, 2) void vit_gemm_kernel(VGemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* lds = reinterpret_cast<bf16_t*>(smem);  // ring: [4 stages][A 128x32 | B 128x32] bf16 = 64 KiB
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  const int nwg = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid / g.tiles_n, tn = bid % g.tiles_n;
  const int m0 = tm * 128, n0 = tn * 128;

  // staging: one wave-instruction = 16 rows x 64 B; wave wv stages rows [wv*32, wv*32+32) of A and of B.
  const int srow = lane >> 2, sslot = lane & 3;
  const bf16_t* asrc[2];
  const bf16_t* bsrc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wv * 32 + i * 16 + srow;
    const int gc = sslot ^ ((r >> 2) & 2);
    int ar = m0 + r;
    ar = ar < g.M ? ar : g.M - 1;
    asrc[i] = g.A + (int64_t)ar * g.lda + gc * 8;
    bsrc[i] = g.W + (int64_t)(n0 + r) * g.K + gc * 8;
  }
  auto stage = [&](int kt) {
    bf16_t* la = lds + (kt % VG_STAGES) * VG_STAGE_ELEMS;
    bf16_t* lb = la + 128 * VG_BK;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      glds16(asrc[i] + kt * VG_BK, la + (wv * 32 + i * 16) * VG_BK + lane * 8);
      glds16(bsrc[i] + kt * VG_BK, lb + (wv * 32 + i * 16) * VG_BK + lane * 8);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / VG_BK;  // >= 4 (checked by the launcher)
  const int l16 = lane & 15, gq = lane >> 4;
  // per-lane fragment offsets inside a stage (swizzled), constant over the K loop
  int offa[4], offb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ra = wm * 64 + i * 16 + l16, rb = wn * 64 + i * 16 + l16;
    offa[i] = ra * VG_BK + ((gq ^ ((ra >> 2) & 2)) * 8);
    offb[i] = 128 * VG_BK + rb * VG_BK + ((gq ^ ((rb >> 2) & 2)) * 8);
  }
  auto load_frags = [&](int kt, bf16x8 (&fa)[4], bf16x8 (&fb)[4]) {
    const bf16_t* ls = lds + (kt % VG_STAGES) * VG_STAGE_ELEMS;
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(ls + offa[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(ls + offb[j]);
  };
  auto mfmas = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) Mma<MODE_BF16>::mma(acc[i][j], fa[i], fb[j]);
  };
  // One pipeline step for K-tile kt whose fragments are already in (ca, cb):
  //   issue the DMA of tile kt+4 (its slot held tile kt-1, whose fragment reads finished an iteration ago),
  //   issue the LDS reads of tile kt+1 into (na, nb), run the 16 MFMAs of tile kt underneath them, then make
  //   tile kt+2 visible: counted vmcnt (tiles kt+3, kt+4 stay in flight) + one barrier.
  auto step = [&](int kt, const bf16x8 (&ca)[4], const bf16x8 (&cb)[4], bf16x8 (&na)[4], bf16x8 (&nb)[4]) {
    if (kt + 4 < nk) stage(kt + 4);
    if (kt + 1 < nk) load_frags(kt + 1, na, nb);
    mfmas(ca, cb);
    const int ahead = nk - 1 - (kt + 2);   // tiles issued beyond kt+2
    // The fragment reads of tile kt+1 had the whole MFMA block to land: retire them here, so that the next step's
    // MFMAs start without an lgkmcnt(0) that would also wait for the reads issued just in front of them.
    // (an empty asm that "uses" the fragment registers: hipcc places its own lgkmcnt wait for them HERE, behind the
    //  MFMAs, and treats them as plain registers afterwards)
    __builtin_amdgcn_sched_barrier(0);   // keep the 16 MFMAs in front of the wait the next line provokes
    asm volatile("" : "+v"(na[0]), "+v"(na[1]), "+v"(na[2]), "+v"(na[3]), "+v"(nb[0]), "+v"(nb[1]), "+v"(nb[2]), "+v"(nb[3]));
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  stage(0);
  stage(1);
  stage(2);
  stage(3);
  bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tiles 0 and 1 landed (this wave's part)
  __builtin_amdgcn_s_barrier();
  load_frags(0, fa0, fb0);
  asm volatile("" : "+v"(fa0[0]), "+v"(fa0[1]), "+v"(fa0[2]), "+v"(fa0[3]), "+v"(fb0[0]), "+v"(fb0[1]), "+v"(fb0[2]), "+v"(fb0[3]));
  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    step(kt, fa0, fb0, fa1, fb1);
    step(kt + 1, fa1, fb1, fa0, fb0);
  }
  if (kt < nk) step(kt, fa0, fb0, fa1, fb1);
  __syncthreads();  // all waves done reading the ring before the epilogue reuses it

  if constexpr (EPI == EPI_BF16_BIAS || EPI == EPI_BF16_BIAS_GELU) {
    constexpr int LDC = 136;  // bf16 elements per staged row (272 B)
    bf16_t* cs = lds;
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
          float v = acc[i][j][r] + bv;
          if constexpr (EPI == EPI_BF16_BIAS_GELU) v = gelu_erf(v);
          else v *= sc;
          cs[(wm * 64 + i * 16 + gq * 4 + r) * LDC + nl] = f2bf(v);
        }
    }
    __syncthreads();
    bf16_t* C = reinterpret_cast<bf16_t*>(g.C);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int q = it * 256 + tid, row = q >> 4, c16 = q & 15;
      if (m0 + row < g.M)
        *reinterpret_cast<uint4*>(C + (int64_t)(m0 + row) * g.ldc + n0 + c16 * 8) =
            *reinterpret_cast<const uint4*>(cs + row * LDC + c16 * 8);
    }
  } else {
    // f32 outputs: two passes of 64 rows through a [64][132] f32 LDS image, then float4 read-modify-write rows
    constexpr int LDF = 132;
    float* cf = reinterpret_cast<float*>(smem);
    float* C = reinterpret_cast<float*>(g.C);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (half) __syncthreads();
      if (wm == half) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) cf[(i * 16 + gq * 4 + r) * LDF + wn * 64 + j * 16 + l16] = acc[i][j][r];
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int q = it * 256 + tid, row = q >> 5, c4 = (q & 31) * 4;
        const int m = m0 + half * 64 + row;
        if (m < g.M) {
          const float4 a = *reinterpret_cast<const float4*>(cf + row * LDF + c4);
          float4* cp = reinterpret_cast<float4*>(C + (int64_t)m * g.ldc + n0 + c4);
          float4 add;
          if constexpr (EPI == EPI_F32_BIAS_RESID) {
            const float4 res = *cp, bb = *reinterpret_cast<const float4*>(g.bias + n0 + c4);
            add = make_float4(res.x + bb.x, res.y + bb.y, res.z + bb.z, res.w + bb.w);
          } else {
            add = *reinterpret_cast<const float4*>(g.table + (int64_t)(m % g.period) * g.N + n0 + c4);
          }
          *cp = make_float4(a.x + add.x, a.y + add.y, a.z + add.z, a.w + add.w);
        }
      }
    }
  }
}

extern "C" int maavss_vit_gemm(const void* A, int lda, const void* W, const float* bias, const float* table, int period,
                               void* C, int ldc, int64_t M, int N, int K, int epilogue, int qscale_cols, float qscale,
                               void* stream) {
  MAAVSS_CHECK_ARG(A && W && C && M > 0, "vit_gemm: bad arguments");
  MAAVSS_CHECK_ARG(N % 128 == 0 && K % VG_BK == 0 && K >= 4 * VG_BK, "vit_gemm: N must be a multiple of 128 and K of 32, K >= 128 (N=%d K=%d)", N, K);
  MAAVSS_CHECK_ARG(lda % 8 == 0 && ldc % 8 == 0, "vit_gemm: leading dimensions must be multiples of 8");
  MAAVSS_CHECK_ARG(epilogue >= 0 && epilogue <= 3, "vit_gemm: unknown epilogue");
  MAAVSS_CHECK_ARG(epilogue == EPI_F32_ROWTABLE ? (table && period > 0) : (bias != nullptr), "vit_gemm: missing bias/table");
  MAAVSS_CHECK_ARG(M < (1LL << 31), "vit_gemm: M too large");
  VGemmArgs g;
  g.A = (const bf16_t*)A; g.W = (const bf16_t*)W; g.bias = bias; g.table = table; g.C = C;
  g.M = (int)M; g.N = N; g.K = K; g.lda = lda; g.ldc = ldc; g.period = period;
  g.qscale_cols = qscale_cols; g.qscale = qscale;
  g.tiles_n = N / 128; g.tiles_m = cdiv(M, 128);
  const dim3 grid(g.tiles_n * g.tiles_m), block(256);
  const size_t smem = VG_STAGES * VG_STAGE_ELEMS * sizeof(bf16_t);   // 80 KiB: two workgroups fill a CU's 160 KiB
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(vit_gemm_kernel<EPI_BF16_BIAS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute(reinterpret_cast<const void*>(vit_gemm_kernel<EPI_BF16_BIAS_GELU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute(reinterpret_cast<const void*>(vit_gemm_kernel<EPI_F32_BIAS_RESID>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute(reinterpret_cast<const void*>(vit_gemm_kernel<EPI_F32_ROWTABLE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  hipStream_t st = (hipStream_t)stream;
  switch (epilogue) {
    case EPI_BF16_BIAS: hipLaunchKernelGGL(vit_gemm_kernel<EPI_BF16_BIAS>, grid, block, smem, st, g); break;
    case EPI_BF16_BIAS_GELU: hipLaunchKernelGGL(vit_gemm_kernel<EPI_BF16_BIAS_GELU>, grid, block, smem, st, g); break;
    case EPI_F32_BIAS_RESID: hipLaunchKernelGGL(vit_gemm_kernel<EPI_F32_BIAS_RESID>, grid, block, smem, st, g); break;
    default: hipLaunchKernelGGL(vit_gemm_kernel<EPI_F32_ROWTABLE>, grid, block, smem, st, g); break;
  }
  MAAVSS_LAUNCH_CHECK("vit_gemm_kernel");
  return MAAVSS_OK;
}
