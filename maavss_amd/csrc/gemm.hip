// Generic f32-in / f32-out GEMM on MFMA:  C[M,N] = act(alpha * op(A)[M,K] . op(B)[N,K]^T) (+ C if beta==1).
//
// Serves the Linear layers of the fusion network and their gradients -- reference
// avse_model_final.py:132-146 (LSTM input projection, fc1, fc2), :203-213/264-268 (a_fc1, v_fc1) and
// what autograd derives from them (dX = dY.W, dW = dY^T.X).  Both operands may be stored
// K-contiguous or K-strided; tiles are staged through registers, converted (bf16 or f32, see mma.h) and
// written to LDS as [row][32 k], so one MFMA inner loop serves NN/NT/TN/TT.  The next K tile's
// global loads are issued before the current tile's MFMAs (register prefetch).
// Small-M problems (M = batch) are weight-streaming, HBM-bound: the launcher swaps the operands so the
// weight is the M side, picks a narrow N tile and splits K across blocks (f32 atomics).
#include "mma.h"

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  int64_t lda, ldb, ldc;
  int M, N, K;
  int transA, transB, transC;
  float alpha;
  int beta, act, split_k, k_per_split;
};

__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == 1) return tanhf(v);
  if (act == 2) return 1.0f / (1.0f + __expf(-v));
  return v;
}

// Loads a [ROWS x 32] operand tile into registers (NV float4 per thread, zero-filled out of range).
template <int ROWS>
__device__ __forceinline__ void tile_load(const float* __restrict__ P, int64_t ld, int trans, int row0, int k0,
                                          int nrows, int kend, float4 (&r)[ROWS / 32]) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int v = 0; v < ROWS / 32; ++v) {
    const int idx = v * 256 + tid;
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!trans) {
      const int row = row0 + (idx >> 3), k = k0 + (idx & 7) * 4;
      if (row < nrows && k < kend) {
        const float* p = P + (int64_t)row * ld + k;
        if (k + 3 < kend && (((uintptr_t)p) & 15) == 0) {
          val = *reinterpret_cast<const float4*>(p);
        } else {
          val.x = p[0];
          if (k + 1 < kend) val.y = p[1];
          if (k + 2 < kend) val.z = p[2];
          if (k + 3 < kend) val.w = p[3];
        }
      }
    } else {
      const int k = k0 + idx / (ROWS / 4), row = row0 + (idx % (ROWS / 4)) * 4;
      if (k < kend && row < nrows) {
        const float* p = P + (int64_t)k * ld + row;
        if (row + 3 < nrows && (((uintptr_t)p) & 15) == 0) {
          val = *reinterpret_cast<const float4*>(p);
        } else {
          val.x = p[0];
          if (row + 1 < nrows) val.y = p[1];
          if (row + 2 < nrows) val.z = p[2];
          if (row + 3 < nrows) val.w = p[3];
        }
      }
    }
    r[v] = val;
  }
}

template <int PRECISE, int ROWS>
__device__ __forceinline__ void tile_store(typename Mma<PRECISE>::elem* S, int trans, const float4 (&r)[ROWS / 32]) {
  using M = Mma<PRECISE>;
  const int tid = threadIdx.x;
#pragma unroll
  for (int v = 0; v < ROWS / 32; ++v) {
    const int idx = v * 256 + tid;
    if (!trans) {
      typename M::elem* d = S + (idx >> 3) * 32 + (idx & 7) * 4;
      d[0] = M::cvt(r[v].x);
      d[1] = M::cvt(r[v].y);
      d[2] = M::cvt(r[v].z);
      d[3] = M::cvt(r[v].w);
    } else {
      const int k = idx / (ROWS / 4), row = (idx % (ROWS / 4)) * 4;
      S[(row + 0) * 32 + k] = M::cvt(r[v].x);
      S[(row + 1) * 32 + k] = M::cvt(r[v].y);
      S[(row + 2) * 32 + k] = M::cvt(r[v].z);
      S[(row + 3) * 32 + k] = M::cvt(r[v].w);
    }
  }
}

template <int PRECISE, int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs g) {
  using M = Mma<PRECISE>;
  constexpr int TM = BM / WGM, TN = BN / WGN, MT = TM / 16, NT = TN / 16;
  __shared__ __attribute__((aligned(16))) typename M::elem As[BM * 32];
  __shared__ __attribute__((aligned(16))) typename M::elem Bs[BN * 32];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wm = wv / WGN, wn = wv % WGN;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float4 ra[BM / 32], rb[BN / 32];
  tile_load<BM>(g.A, g.lda, g.transA, m0, kbeg, g.M, kend, ra);
  tile_load<BN>(g.B, g.ldb, g.transB, n0, kbeg, g.N, kend, rb);
  for (int k0 = kbeg; k0 < kend; k0 += 32) {
    tile_store<PRECISE, BM>(As, g.transA, ra);
    tile_store<PRECISE, BN>(Bs, g.transB, rb);
    __syncthreads();
    if (k0 + 32 < kend) {
      tile_load<BM>(g.A, g.lda, g.transA, m0, k0 + 32, g.M, kend, ra);
      tile_load<BN>(g.B, g.ldb, g.transB, n0, k0 + 32, g.N, kend, rb);
    }
    typename M::frag fa[MT], fb[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[i] = M::load(As + (wm * TM + i * 16 + (lane & 15)) * 32 + (lane >> 4) * 8);
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[j] = M::load(Bs + (wn * TN + j * 16 + (lane & 15)) * 32 + (lane >> 4) * 8);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) M::mma(acc[i][j], fa[i], fb[j]);
    __syncthreads();
  }
  // epilogue
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * TM + i * 16 + (lane >> 4) * 4 + r;
        const int n = n0 + wn * TN + j * 16 + (lane & 15);
        if (m < g.M && n < g.N) {
          float* c = g.C + (g.transC ? (int64_t)n * g.ldc + m : (int64_t)m * g.ldc + n);
          float v = acc[i][j][r] * g.alpha;
          if (g.split_k > 1) {
            atomicAdd(c, v);
          } else {
            if (g.beta) v += *c;
            *c = apply_act(v, g.act);
          }
        }
      }
}

__global__ void gemm_zero_kernel(float* C, int64_t ldc, int rows, int cols) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)rows * cols;
       i += (int64_t)gridDim.x * blockDim.x)
    C[(i / cols) * ldc + (i % cols)] = 0.f;
}
__global__ void gemm_act_kernel(float* C, int64_t ldc, int rows, int cols, int act) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)rows * cols;
       i += (int64_t)gridDim.x * blockDim.x) {
    float* c = C + (i / cols) * ldc + (i % cols);
    *c = apply_act(*c, act);
  }
}

template <int PRECISE>
static void launch_gemm(GemmArgs g, int cfg, hipStream_t st) {
  if (cfg == 0) {
    dim3 grid(cdiv(g.N, 32), cdiv(g.M, 128), g.split_k);
    hipLaunchKernelGGL((gemm_kernel<PRECISE, 128, 32, 4, 1>), grid, dim3(256), 0, st, g);
  } else if (cfg == 1) {
    dim3 grid(cdiv(g.N, 64), cdiv(g.M, 64), g.split_k);
    hipLaunchKernelGGL((gemm_kernel<PRECISE, 64, 64, 2, 2>), grid, dim3(256), 0, st, g);
  } else {
    dim3 grid(cdiv(g.N, 128), cdiv(g.M, 128), g.split_k);
    hipLaunchKernelGGL((gemm_kernel<PRECISE, 128, 128, 2, 2>), grid, dim3(256), 0, st, g);
  }
}

int maavss_linear_skinny_try(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C, int64_t ldc,
                             int transC, int64_t M, int64_t N, int64_t K, float alpha, int beta, int act, hipStream_t st, int* taken);

extern "C" int maavss_gemm_f32(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB,
                               float* C, int64_t ldc, int transC, int64_t M, int64_t N, int64_t K, float alpha,
                               int beta, int act, int split_k, int precise, void* stream) {
  MAAVSS_CHECK_ARG(A && B && C, "gemm: null pointer");
  MAAVSS_CHECK_ARG(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
  MAAVSS_CHECK_ARG(M < (1LL << 31) && N < (1LL << 31) && K < (1LL << 31), "gemm: dimension too large");
  MAAVSS_CHECK_ARG(beta == 0 || beta == 1, "gemm: beta must be 0 or 1");
  MAAVSS_CHECK_ARG(precise >= 0 && precise <= 2, "gemm: mode must be 0 (bf16), 1 (f32) or 2 (f16)");
  MAAVSS_CHECK_ARG(act >= 0 && act <= 2, "gemm: act must be 0 (none), 1 (tanh) or 2 (sigmoid)");
  hipStream_t st = (hipStream_t)stream;
  if (precise == MODE_F32 && split_k <= 0) {   // the Linear layers at M = batch: dedicated weight-streaming forms (linear_skinny.hip)
    int taken = 0;
    const int rc = maavss_linear_skinny_try(A, lda, transA, B, ldb, transB, C, ldc, transC, M, N, K, alpha, beta, act, st, &taken);
    if (rc != MAAVSS_OK || taken) return rc;
  }
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.transA = transA; g.transB = transB; g.transC = transC;
  g.alpha = alpha; g.beta = beta; g.act = act;
  if (g.M <= 32 && g.N > 32) {  // skinny M: make the long operand the M side (C^T = B . A^T)
    GemmArgs s = g;
    s.A = g.B; s.lda = g.ldb; s.transA = g.transB;
    s.B = g.A; s.ldb = g.lda; s.transB = g.transA;
    s.M = g.N; s.N = g.M; s.transC = !g.transC;
    g = s;
  }
  int cfg;
  if (g.N <= 32) cfg = 0;
  else if ((int64_t)g.M * g.N >= 128LL * 128 * 512) cfg = 2;
  else cfg = 1;
  const int bm = cfg == 1 ? 64 : 128, bn = cfg == 0 ? 32 : (cfg == 1 ? 64 : 128);
  const int64_t blocks = (int64_t)cdiv(g.M, bm) * cdiv(g.N, bn);
  if (split_k <= 0) {  // auto: fill the chip when the output grid alone cannot (never in deterministic mode: split-K sums with atomics)
    split_k = 1;
    if (blocks < 512 && g.K >= 512 && !maavss_deterministic_flag()) {
      split_k = (int)((1024 + blocks - 1) / blocks);
      const int maxs = g.K / 128;
      if (split_k > maxs) split_k = maxs;
      if (split_k < 1) split_k = 1;
    }
  }
  int kps = cdiv(cdiv(g.K, split_k), 32) * 32;
  split_k = cdiv(g.K, kps);
  g.split_k = split_k;
  g.k_per_split = kps;
  const int rows = g.transC ? g.N : g.M, cols = g.transC ? g.M : g.N;
  if (split_k > 1 && !beta) {
    hipLaunchKernelGGL(gemm_zero_kernel, dim3(min(2048, cdiv((int64_t)rows * cols, 256))), dim3(256), 0, st, C, ldc,
                       rows, cols);
  }
  if (precise == MODE_F32) launch_gemm<MODE_F32>(g, cfg, st);
  else if (precise == MODE_F16) launch_gemm<MODE_F16>(g, cfg, st);
  else launch_gemm<MODE_BF16>(g, cfg, st);
  MAAVSS_LAUNCH_CHECK("gemm_kernel");
  if (split_k > 1 && act != 0) {
    hipLaunchKernelGGL(gemm_act_kernel, dim3(min(2048, cdiv((int64_t)rows * cols, 256))), dim3(256), 0, st, C, ldc,
                       rows, cols, act);
    MAAVSS_LAUNCH_CHECK("gemm_act_kernel");
  }
  return MAAVSS_OK;
}
