// Linear layers at M = batch <= 32 (fc1 8192 -> 4096, v_fc1 512 -> W^2, fc2, a_fc1 of avse_model_final.py:141-146,203-213 and the
// gradients autograd derives from them): pure weight streaming -- 103 / 134 MB of f32 weights read (forward, input gradient) or
// written (weight gradient) per call against 32 rows of activations.  The generic tiled kernel (gemm.hip: one 32-deep K step per
// pair of workgroup barriers, single-buffered LDS) streams them at 1.0-1.4 TB/s; these three forms keep the weight on a straight
// float4 path HBM -> registers -> MFMA and everything else in LDS:
//   fwd : out[m][n] (+)= sum_k X[m][k] W[n][k]     a wave owns 16 weight rows x a K slice, X slice in LDS, split-K by atomics
//   dx  : out[m][k] (+)= sum_n dY[m][n] W[n][k]    a wave owns 64 output columns x an N slice, dY slice (transposed) in LDS
//   dw  : dW[n][k]  (+)= sum_m dY[m][n] X[m][k]    write-bound: X fragments in registers, dY slice in LDS, float4 stores
// Arithmetic: v_mfma_f32_16x16x4_f32 (exact f32 products and sums, like the generic kernel's precise mode).  A lane feeds the
// four MFMAs of a 16-deep step with the four elements of ONE float4 -- MFMA e takes element e of every lane, i.e. the k set
// {e, 4 + e, 8 + e, 12 + e} -- so operands are loaded 16 bytes per lane and no shuffle is needed (any split of K over the MFMAs
// is valid as long as both operands use the same one).
#include "mma.h"

#define LS_KS 512          // fwd: K slice per workgroup (X slice [32][LS_KS] in LDS)
#define LS_NS 512          // dx: N slice per workgroup (64 rows per wave)
#define LS_XPAD 4

struct LinArgs {
  const float* act;      // X or dY [Mb][..]
  const float* w;        // W [N][K] (fwd, dx) -- or X for dw
  float* out;
  int64_t ld_act, ld_w, ld_out;
  int Mb, N, K;
  float alpha;
  int beta, actfn, atomic;   // atomic: 0 single slice, 1 split over slices with f32 atomics, 2 deterministic partials in `part`
  float* part;               // atomic == 2: [slices][32][cols]
};

__device__ __forceinline__ float ls_act(float v, int a) {
  if (a == 1) return tanhf(v);
  if (a == 2) return 1.0f / (1.0f + __expf(-v));
  return v;
}
__device__ __forceinline__ void ls_mfma(f32x4& acc, float a, float b) { acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0); }

// ---- fwd: grid (ceil(N / 128), ceil(K / LS_KS)), 512 threads
__global__ __launch_bounds__(512) void linear_fwd_skinny_kernel(LinArgs g) {
  __shared__ __attribute__((aligned(16))) float xs[32][LS_KS + LS_XPAD];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l16 = lane & 15, gq = lane >> 4;
  const int k0 = blockIdx.y * LS_KS, kn = min(LS_KS, g.K - k0);       // kn is a multiple of 16 (checked by the launcher)
  for (int i = tid; i < 32 * (LS_KS / 4); i += 512) {
    const int m = i / (LS_KS / 4), c = (i % (LS_KS / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < g.Mb && c < kn) v = *reinterpret_cast<const float4*>(g.act + (int64_t)m * g.ld_act + k0 + c);
    *reinterpret_cast<float4*>(&xs[m][c]) = v;
  }
  const int n0 = blockIdx.x * 128 + wv * 16;
  const int row = min(n0 + l16, g.N - 1);
  const float* wp = g.w + (int64_t)row * g.ld_w + k0 + 4 * gq;
  __syncthreads();
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < kn; s0 += 128) {
    float4 wb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int kk = s0 + 16 * u;
      wb[u] = kk < kn ? *reinterpret_cast<const float4*>(wp + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int kk = min(s0 + 16 * u, LS_KS - 16) + 4 * gq;
      const float4 x0 = *reinterpret_cast<const float4*>(&xs[l16][kk]), x1 = *reinterpret_cast<const float4*>(&xs[16 + l16][kk]);
      ls_mfma(acc0, x0.x, wb[u].x); ls_mfma(acc1, x1.x, wb[u].x);
      ls_mfma(acc0, x0.y, wb[u].y); ls_mfma(acc1, x1.y, wb[u].y);
      ls_mfma(acc0, x0.z, wb[u].z); ls_mfma(acc1, x1.z, wb[u].z);
      ls_mfma(acc0, x0.w, wb[u].w); ls_mfma(acc1, x1.w, wb[u].w);
    }
  }
  // D[i = m = 4 gq + r (+16)][j = weight row l16]: the 16 lanes of a row group touch 16 consecutive output columns (64 B) -- the
  // split-K atomics and the stores coalesce
  const int n = n0 + l16;
  if (n < g.N) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4 a = h ? acc1 : acc0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * h + 4 * gq + r;
        if (m >= g.Mb) continue;
        float* c = g.out + (int64_t)m * g.ld_out + n;
        float v = a[r] * g.alpha;
        if (g.atomic == 2) {                 // deterministic: this K slice's partial sum, added in slice order by linear_reduce_kernel
          g.part[((int64_t)blockIdx.y * 32 + m) * g.N + n] = v;
        } else if (g.atomic) {
          atomicAdd(c, v);
        } else {
          if (g.beta) v += *c;
          *c = ls_act(v, g.actfn);
        }
      }
    }
  }
}

// ---- dx: out[m][k] (+)= sum_n dY[m][n] W[n][k].  grid (ceil(K / 64), ceil(N / LS_NS)), 512 threads.  The eight waves share 64
// output columns and split the workgroup's N slice (LS_NS / 8 rows each); their partial sums meet in LDS, so the number of
// atomics per output element is the number of N slices, not of waves.
__global__ __launch_bounds__(512) void linear_dx_skinny_kernel(LinArgs g) {
  constexpr int NW = LS_NS / 8;
  __shared__ __attribute__((aligned(16))) float lds[LS_NS * 32];   // dY slice [n][m]; afterwards the partial sums [wave][m][64]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l16 = lane & 15, gq = lane >> 4;
  const int n0 = blockIdx.y * LS_NS, nn = min(LS_NS, g.N - n0);
  for (int i = tid; i < 32 * LS_NS; i += 512) {
    const int m = i / LS_NS, n = i % LS_NS;
    lds[n * 32 + m] = (m < g.Mb && n < nn) ? g.act[(int64_t)m * g.ld_act + n0 + n] : 0.f;
  }
  const int kw0 = blockIdx.x * 64;
  const int kcol = min(kw0 + 4 * l16, g.K - 4);     // K is a multiple of 4 (launcher); clamped lanes are masked at the store
  const int nb = wv * NW;                            // this wave's rows of the slice
  const float* wp = g.w + (int64_t)(n0 + nb) * g.ld_w + kcol;
  __syncthreads();
  f32x4 acc[4][2];
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e][0] = acc[e][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int s0 = 0; s0 < NW; s0 += 32) {
    float4 wb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = s0 + 4 * u + gq;
      wb[u] = nb + n < nn ? *reinterpret_cast<const float4*>(wp + (int64_t)n * g.ld_w) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = nb + s0 + 4 * u + gq;           // < LS_NS; rows >= nn hold zeros
      const float d0 = lds[n * 32 + l16], d1 = lds[n * 32 + 16 + l16];
      ls_mfma(acc[0][0], d0, wb[u].x); ls_mfma(acc[0][1], d1, wb[u].x);
      ls_mfma(acc[1][0], d0, wb[u].y); ls_mfma(acc[1][1], d1, wb[u].y);
      ls_mfma(acc[2][0], d0, wb[u].z); ls_mfma(acc[2][1], d1, wb[u].z);
      ls_mfma(acc[3][0], d0, wb[u].w); ls_mfma(acc[3][1], d1, wb[u].w);
    }
  }
  // MFMA e: D[i = m = 4 gq + r (+16)][j = l16] <-> output column kw0 + 4 l16 + e
  __syncthreads();                                   // everybody is done with the dY slice
  float* red = lds + wv * (32 * 64);
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<float4*>(red + (16 * h + 4 * gq + r) * 64 + 4 * l16) = make_float4(acc[0][h][r], acc[1][h][r], acc[2][h][r], acc[3][h][r]);
  __syncthreads();
  {
    const int m = tid >> 4, c4 = (tid & 15) * 4, k = kw0 + c4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const float4 p = *reinterpret_cast<const float4*>(lds + w * (32 * 64) + m * 64 + c4);
      v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    if (m < g.Mb && k < g.K) {
      float* c = g.out + (int64_t)m * g.ld_out + k;
      v.x *= g.alpha; v.y *= g.alpha; v.z *= g.alpha; v.w *= g.alpha;
      if (g.atomic == 2) {
        *reinterpret_cast<float4*>(g.part + ((int64_t)blockIdx.y * 32 + m) * g.K + k) = v;
      } else if (g.atomic) {
        atomicAdd(c, v.x); atomicAdd(c + 1, v.y); atomicAdd(c + 2, v.z); atomicAdd(c + 3, v.w);
      } else {
        if (g.beta) { const float4 o = *reinterpret_cast<const float4*>(c); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *reinterpret_cast<float4*>(c) = make_float4(ls_act(v.x, g.actfn), ls_act(v.y, g.actfn), ls_act(v.z, g.actfn), ls_act(v.w, g.actfn));
      }
    }
  }
}

// ---- dw: dW[n][k] (+)= sum_m dY[m][n] X[m][k], m < Mb <= 32.  grid (ceil(K / 256), ceil(N / 256)), 256 threads; a wave keeps the
// X fragments of 64 columns in registers and walks the 256 rows of the workgroup's dY slice
__global__ __launch_bounds__(256) void linear_dw_skinny_kernel(LinArgs g) {   // act = dY [Mb][N], w = X [Mb][K], out = dW [N][K]
  __shared__ float dys[32][256 + 16];               // row stride = 16 banks mod 64: the four rows of a fragment read do not collide
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l16 = lane & 15, gq = lane >> 4;
  const int n0 = blockIdx.y * 256;
  for (int i = tid; i < 32 * 256; i += 256) {
    const int m = i >> 8, n = i & 255;
    dys[m][n] = (m < g.Mb && n0 + n < g.N) ? g.act[(int64_t)m * g.ld_act + n0 + n] : 0.f;
  }
  const int k0 = blockIdx.x * 256 + wv * 64;
  float xa[4][8];                                    // A operand: lane (row i = l16 <-> column k0 + 16 t + l16, member gq <-> m = 4 s + gq)
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int m = 4 * s + gq, k = k0 + 16 * t + l16;
      xa[t][s] = (m < g.Mb && k < g.K) ? g.w[(int64_t)m * g.ld_w + k] : 0.f;
    }
  __syncthreads();
  for (int nt = 0; nt < 16; ++nt) {
    const int n = n0 + 16 * nt + l16;
    float db[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) db[s] = dys[4 * s + gq][16 * nt + l16];
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 8; ++s) ls_mfma(acc[t], xa[t][s], db[s]);
    }
    if (n < g.N) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int k = k0 + 16 * t + 4 * gq;          // D[i = 4 gq + r][j = n]: four consecutive columns of row n
        if (k >= g.K) continue;
        float* c = g.out + (int64_t)n * g.ld_out + k;
        float4 v = make_float4(acc[t][0] * g.alpha, acc[t][1] * g.alpha, acc[t][2] * g.alpha, acc[t][3] * g.alpha);
        if (g.beta) { const float4 o = *reinterpret_cast<const float4*>(c); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        *reinterpret_cast<float4*>(c) = v;
      }
    }
  }
}

__global__ void linear_zero_kernel(float* C, int64_t ldc, int rows, int cols) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)rows * cols; i += (int64_t)gridDim.x * blockDim.x)
    C[(i / cols) * ldc + (i % cols)] = 0.f;
}
__global__ void linear_act_kernel(float* C, int64_t ldc, int rows, int cols, int act) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)rows * cols; i += (int64_t)gridDim.x * blockDim.x) {
    float* c = C + (i / cols) * ldc + (i % cols);
    *c = ls_act(*c, act);
  }
}

// deterministic split: C[m][n] = act(sum over slices in order (+ C when beta))
__global__ void linear_reduce_kernel(const float* __restrict__ part, int slices, float* C, int64_t ldc, int rows, int cols, int beta, int act) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (int64_t)rows * cols; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / cols), n = (int)(i % cols);
    float* c = C + (int64_t)m * ldc + n;
    float v = beta ? *c : 0.f;
    for (int s = 0; s < slices; ++s) v += part[((int64_t)s * 32 + m) * cols + n];
    *c = ls_act(v, act);
  }
}

static int ls_blocks(int64_t elems) { const int b = cdiv(elems, 256); return b < 2048 ? b : 2048; }
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Called by maavss_gemm_f32 (gemm.hip) for its exact-f32 mode.  *taken = 1 if one of the three forms took the problem, 0 if the
// generic kernel has to (shape / alignment outside what they cover); returns a status code.
int maavss_linear_skinny_try(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C, int64_t ldc,
                             int transC, int64_t M, int64_t N, int64_t K, float alpha, int beta, int act, hipStream_t st, int* taken) {
  *taken = 0;
  if (transC) return MAAVSS_OK;
  LinArgs g;
  g.alpha = alpha; g.beta = beta; g.actfn = act; g.atomic = 0; g.part = nullptr;
  // deterministic mode: multi-slice cases are taken only when the caller provided scratch for the partial sums
  const bool det = maavss_deterministic_flag() != 0;
  int64_t ws_floats = 0;
  float* det_ws = maavss_deterministic_ws(&ws_floats, (void*)st);
  auto det_ok = [&](int64_t reduce_len, int slice_len, int64_t cols) { return !det || reduce_len <= slice_len || (det_ws && (int64_t)cdiv(reduce_len, slice_len) * 32 * cols <= ws_floats); };
  if (!transA && !transB && det_ok(K, LS_KS, N) && M <= 32 && N >= 512 && N % 4 == 0 && K % 16 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 &&
      aligned16(A) && aligned16(B) && aligned16(C)) {
    // fwd: C[M][N] = A[M][K] . B[N][K]^T
    g.act = A; g.ld_act = lda; g.w = B; g.ld_w = ldb; g.out = C; g.ld_out = ldc; g.Mb = (int)M; g.N = (int)N; g.K = (int)K;
    const int slices = cdiv(K, LS_KS);
    g.atomic = slices > 1 ? (det ? 2 : 1) : 0;
    g.part = det_ws;
    if (g.atomic == 1 && !beta) hipLaunchKernelGGL(linear_zero_kernel, dim3(ls_blocks(M * N)), dim3(256), 0, st, C, ldc, (int)M, (int)N);
    hipLaunchKernelGGL(linear_fwd_skinny_kernel, dim3(cdiv(N, 128), slices), dim3(512), 0, st, g);
    if (g.atomic == 1 && act) hipLaunchKernelGGL(linear_act_kernel, dim3(ls_blocks(M * N)), dim3(256), 0, st, C, ldc, (int)M, (int)N, act);
    if (g.atomic == 2) hipLaunchKernelGGL(linear_reduce_kernel, dim3(ls_blocks(M * N)), dim3(256), 0, st, det_ws, slices, C, ldc, (int)M, (int)N, beta, act);
    MAAVSS_LAUNCH_CHECK("linear_fwd_skinny_kernel");
    *taken = 1;
    return MAAVSS_OK;
  }
  if (!transA && transB && det_ok(K, LS_NS, N) && M <= 32 && N >= 256 && N % 4 == 0 && K >= LS_NS && ldb % 4 == 0 && ldc % 4 == 0 && aligned16(B) && aligned16(C)) {
    // dx: C[M][N] = A[M][K] . B[K][N]   (B = the weight [K rows][N columns]; K is the reduction)
    g.act = A; g.ld_act = lda; g.w = B; g.ld_w = ldb; g.out = C; g.ld_out = ldc; g.Mb = (int)M; g.N = (int)K; g.K = (int)N;
    const int slices = cdiv(K, LS_NS);
    g.atomic = slices > 1 ? (det ? 2 : 1) : 0;
    g.part = det_ws;
    if (g.atomic == 1 && !beta) hipLaunchKernelGGL(linear_zero_kernel, dim3(ls_blocks(M * N)), dim3(256), 0, st, C, ldc, (int)M, (int)N);
    hipLaunchKernelGGL(linear_dx_skinny_kernel, dim3(cdiv(N, 64), slices), dim3(512), 0, st, g);
    if (g.atomic == 1 && act) hipLaunchKernelGGL(linear_act_kernel, dim3(ls_blocks(M * N)), dim3(256), 0, st, C, ldc, (int)M, (int)N, act);
    if (g.atomic == 2) hipLaunchKernelGGL(linear_reduce_kernel, dim3(ls_blocks(M * N)), dim3(256), 0, st, det_ws, slices, C, ldc, (int)M, (int)N, beta, act);
    MAAVSS_LAUNCH_CHECK("linear_dx_skinny_kernel");
    *taken = 1;
    return MAAVSS_OK;
  }
  if (transA && transB && K <= 32 && M >= 256 && N >= 256 && N % 4 == 0 && ldc % 4 == 0 && aligned16(C) && act == 0) {
    // dw: C[M][N] = A[K][M]^T . B[K][N]   (A = dY [batch][M], B = X [batch][N])
    g.act = A; g.ld_act = lda; g.w = B; g.ld_w = ldb; g.out = C; g.ld_out = ldc; g.Mb = (int)K; g.N = (int)M; g.K = (int)N;
    hipLaunchKernelGGL(linear_dw_skinny_kernel, dim3(cdiv(N, 256), cdiv(M, 256)), dim3(256), 0, st, g);
    MAAVSS_LAUNCH_CHECK("linear_dw_skinny_kernel");
    *taken = 1;
    return MAAVSS_OK;
  }
  return MAAVSS_OK;
}
