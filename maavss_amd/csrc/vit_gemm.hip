// K1/K3: the dense layers of the DINO ViT-S/8 (patch projection, qkv, proj, MLP fc1/fc2) as ONE bf16 MFMA GEMM
// with fused epilogues.  Replaces the Linear/Conv2d calls inside dino's VisionTransformer that the reference
// reaches through video_attention.py:52 (get_last_selfattention); the module is external to the reference
// (empty submodule), see oracle/vit_ref_cpu.py for the restated architecture.
//
//   C[M,N] = epilogue( A[M,K] (bf16, row-major) . W[N,K]^T (bf16, torch Linear layout) )
//
// 128x128 output tile per 256-thread workgroup (2x2 waves, 64x64 per wave = 4x4 MFMA 16x16x32 tiles),
// BK = 64.  Operand tiles go global -> LDS directly (global_load_lds, 16 B per lane), double-buffered; the
// LDS image is lane-linear, so the bank-conflict swizzle (16-B chunk ^= row & 7) is applied to the per-lane
// SOURCE address and again on the ds_read_b128 side.  Blocks are remapped so that the N-tiles of one M-panel
// run on the same XCD (A panel re-read from that XCD's L2, not from HBM).
// Epilogues: +bias (and 1/8 on the q third) -> bf16 | +bias, exact-erf GELU -> bf16 | +bias +residual -> f32
// (in place on the residual stream) | + periodic row table (conv bias / cls token + position embedding) -> f32.
// bf16 outputs are staged through LDS and stored as full 256-byte rows.
#include "mma.h"

#define EPI_BF16_BIAS 0
#define EPI_BF16_BIAS_GELU 1
#define EPI_F32_BIAS_RESID 2
#define EPI_F32_ROWTABLE 3

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

template <int EPI>
__global__ __launch_bounds__(256) void vit_gemm_kernel(VGemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [2 buffers][A 128x64 | B 128x64] bf16 = 2 * 32 KiB ; epilogue reuses it as a [128][136] bf16 image
  bf16_t* lds = reinterpret_cast<bf16_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wm = wv >> 1, wn = wv & 1;
  // XCD-aware remap: consecutive ids of one XCD walk the N tiles of the same M panel.
  const int nwg = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = bid / g.tiles_n, tn = bid % g.tiles_n;
  const int m0 = tm * 128, n0 = tn * 128;

  // staging: one wave-instruction = 8 rows x 128 B; wave wv stages rows [wv*32, wv*32+32) of A and of B.
  const int srow = lane >> 3, schunk = lane & 7;
  auto stage = [&](int kt, int buf) {
    bf16_t* la = lds + buf * (2 * 128 * 64);
    bf16_t* lb = la + 128 * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = wv * 32 + i * 8 + srow;
      const int gc = schunk ^ (r & 7);
      int ar = m0 + r;
      ar = ar < g.M ? ar : g.M - 1;
      glds16(g.A + (int64_t)ar * g.lda + kt * 64 + gc * 8, la + (wv * 32 + i * 8) * 64 + lane * 8);
      glds16(g.W + (int64_t)(n0 + r) * g.K + kt * 64 + gc * 8, lb + (wv * 32 + i * 8) * 64 + lane * 8);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / 64;
  stage(0, 0);
  __syncthreads();  // drains vmcnt(0) too
  const int l16 = lane & 15, gq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
    const bf16_t* la = lds + buf * (2 * 128 * 64);
    const bf16_t* lb = la + 128 * 64;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = wm * 64 + i * 16 + l16;
        fa[i] = *reinterpret_cast<const bf16x8*>(la + r * 64 + (((s * 4 + gq) ^ (r & 7)) * 8));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wn * 64 + j * 16 + l16;
        fb[j] = *reinterpret_cast<const bf16x8*>(lb + r * 64 + (((s * 4 + gq) ^ (r & 7)) * 8));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Mma<MODE_BF16>::mma(acc[i][j], fa[i], fb[j]);
    }
    __syncthreads();
  }

  if constexpr (EPI == EPI_BF16_BIAS || EPI == EPI_BF16_BIAS_GELU) {
    constexpr int LDC = 136;  // bf16 elements per staged row (272 B: 16-B aligned, breaks the 256-B bank period)
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
          if constexpr (EPI == EPI_BF16_BIAS_GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
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
    float* C = reinterpret_cast<float*>(g.C);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 64 + i * 16 + gq * 4 + r;
        if (m < g.M) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + l16;
            float* c = C + (int64_t)m * g.ldc + n;
            if constexpr (EPI == EPI_F32_BIAS_RESID) *c = *c + acc[i][j][r] + g.bias[n];
            else *c = acc[i][j][r] + g.table[(int64_t)(m % g.period) * g.N + n];
          }
        }
      }
  }
}

extern "C" int maavss_vit_gemm(const void* A, int lda, const void* W, const float* bias, const float* table, int period,
                               void* C, int ldc, int64_t M, int N, int K, int epilogue, int qscale_cols, float qscale,
                               void* stream) {
  MAAVSS_CHECK_ARG(A && W && C && M > 0, "vit_gemm: bad arguments");
  MAAVSS_CHECK_ARG(N % 128 == 0 && K % 64 == 0 && K >= 64, "vit_gemm: N must be a multiple of 128 and K of 64 (N=%d K=%d)", N, K);
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
  const size_t smem = 2 * 2 * 128 * 64 * sizeof(bf16_t);
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
