// K10 (convolution part): the STFT encoder's Conv2d(k=(3,9), stride (sh,sw) in {1,2}^2, pad (1,pw), bias=False)
// layers (reference avse_model_final.py:98-102), forward / input gradient / weight gradient.
// 2 -> 4 -> 8 -> 16 (-> 16) channels on at most [128 x 257] maps: a few MFLOP per clip, direct kernels, latency-bound.
// Round 2: one thread per POSITION with all output (input) channels in registers -- every loaded value feeds C FMAs against
// wave-uniform weights -- and a weight gradient whose blocks own a (ci, kh) row of 9 taps x all C_out; the first version
// (one thread per output ELEMENT, one block per (co, ci) pair) re-read each input C times and cost 1.05 ms per step for
// 0.2 GFLOP.
// Input layout 0 = NCHW (the network input x_stft [B,2,T_a,F]) or 1 = NHWC; outputs are NHWC.
// Weights stay in the reference layout [Co][Ci][3][9].
#include "common.h"

struct C2Geom {
  int B, Ci, H, W, Co, Ho, Wo, sh, sw, pw, in_layout;
};

__device__ __forceinline__ int64_t in_index(const C2Geom& g, int b, int ci, int iy, int ix) {
  return g.in_layout ? (((int64_t)b * g.H + iy) * g.W + ix) * g.Ci + ci : (((int64_t)b * g.Ci + ci) * g.H + iy) * g.W + ix;
}

__global__ __launch_bounds__(256) void conv2d_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         float* __restrict__ y, C2Geom g) {
  const int64_t total = (int64_t)g.B * g.Ho * g.Wo * g.Co;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int co = (int)(i % g.Co);
    const int64_t pos = i / g.Co;
    const int ox = (int)(pos % g.Wo), oy = (int)((pos / g.Wo) % g.Ho), b = (int)(pos / ((int64_t)g.Wo * g.Ho));
    float acc = 0.f;
    for (int ci = 0; ci < g.Ci; ++ci)
      for (int kh = 0; kh < 3; ++kh) {
        const int iy = oy * g.sh + kh - 1;
        if (iy < 0 || iy >= g.H) continue;
        const float* wp = w + (((int64_t)co * g.Ci + ci) * 3 + kh) * 9;
#pragma unroll
        for (int kw = 0; kw < 9; ++kw) {
          const int ix = ox * g.sw + kw - g.pw;
          if (ix >= 0 && ix < g.W) acc = fmaf(x[in_index(g, b, ci, iy, ix)], wp[kw], acc);
        }
      }
    y[i] = acc;
  }
}

template <int CO>
__global__ __launch_bounds__(256) void conv2d_fwd_pos_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             float* __restrict__ y, C2Geom g) {
  // Weights in LDS as [ci][kh][kw][CO]: the CO weights of a tap are CO / 4 broadcast 16-byte reads.  Straight from global memory they were CO
  // scalar loads per tap -- 1.4 scalar-memory instructions per vector instruction, a quarter of the waves' cycles on the scalar unit
  // (profiles/r3_f_scalar_pmc.json).  Same FMA order: results unchanged.
  __shared__ __attribute__((aligned(16))) float wl[16 * 27 * CO];
  for (int i = threadIdx.x; i < g.Ci * 27 * CO; i += 256) {
    const int c = i % CO, t = i / CO;                  // t = ci * 27 + kh * 9 + kw
    wl[i] = w[(int64_t)c * g.Ci * 27 + t];
  }
  __syncthreads();
  const int64_t npos = (int64_t)g.B * g.Ho * g.Wo;
  for (int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x; pos < npos; pos += (int64_t)gridDim.x * 256) {
    const int ox = (int)(pos % g.Wo), oy = (int)((pos / g.Wo) % g.Ho), b = (int)(pos / ((int64_t)g.Wo * g.Ho));
    float acc[CO];
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[c] = 0.f;
    for (int ci = 0; ci < g.Ci; ++ci)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int iy = oy * g.sh + kh - 1;
        const bool oky = iy >= 0 && iy < g.H;
#pragma unroll
        for (int kw = 0; kw < 9; ++kw) {
          const int ix = ox * g.sw + kw - g.pw;
          float v = 0.f;
          if (oky && ix >= 0 && ix < g.W) v = x[in_index(g, b, ci, iy, ix)];
          const float4* wp = reinterpret_cast<const float4*>(wl + ((ci * 3 + kh) * 9 + kw) * CO);
#pragma unroll
          for (int c = 0; c < CO; c += 4) {
            const float4 w4 = wp[c / 4];
            acc[c] = fmaf(v, w4.x, acc[c]); acc[c + 1] = fmaf(v, w4.y, acc[c + 1]);
            acc[c + 2] = fmaf(v, w4.z, acc[c + 2]); acc[c + 3] = fmaf(v, w4.w, acc[c + 3]);
          }
        }
      }
    float4* yp = reinterpret_cast<float4*>(y + pos * CO);
#pragma unroll
    for (int c = 0; c < CO; c += 4) yp[c / 4] = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
  }
}

// one thread per INPUT position, all CI gradients in registers; dy rows read as float4
template <int CI>
__global__ __launch_bounds__(256) void conv2d_dgrad_pos_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                               float* __restrict__ dx, C2Geom g) {
  // weights in LDS as [kh][kw][co][CI] (see conv2d_fwd_pos_kernel): the CI weights of (tap, co) are CI / 4 broadcast reads (CI = 2: one 8-byte read)
  __shared__ __attribute__((aligned(16))) float wl[27 * 16 * CI];
  for (int i = threadIdx.x; i < 27 * g.Co * CI; i += 256) {
    const int c = i % CI, co = (i / CI) % g.Co, t = i / (CI * g.Co);      // t = kh * 9 + kw
    wl[i] = w[((int64_t)co * CI + c) * 27 + t];
  }
  __syncthreads();
  const int64_t npos = (int64_t)g.B * g.H * g.W;
  for (int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x; pos < npos; pos += (int64_t)gridDim.x * 256) {
    const int ix = (int)(pos % g.W), iy = (int)((pos / g.W) % g.H), b = (int)(pos / ((int64_t)g.W * g.H));
    float acc[CI];
#pragma unroll
    for (int c = 0; c < CI; ++c) acc[c] = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ty = iy + 1 - kh;
      if (ty < 0 || ty % g.sh != 0) continue;
      const int oy = ty / g.sh;
      if (oy >= g.Ho) continue;
#pragma unroll
      for (int kw = 0; kw < 9; ++kw) {
        const int tx = ix + g.pw - kw;
        if (tx < 0 || tx % g.sw != 0) continue;
        const int ox = tx / g.sw;
        if (ox >= g.Wo) continue;
        const float4* dp = reinterpret_cast<const float4*>(dy + (((int64_t)b * g.Ho + oy) * g.Wo + ox) * g.Co);
        for (int c4 = 0; c4 < g.Co / 4; ++c4) {
          const float4 d = dp[c4];
          const float dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float* wp = wl + ((kh * 9 + kw) * g.Co + c4 * 4 + e) * CI;
            if constexpr (CI % 4 == 0) {
#pragma unroll
              for (int c = 0; c < CI; c += 4) {
                const float4 w4 = *reinterpret_cast<const float4*>(wp + c);
                acc[c] = fmaf(dd[e], w4.x, acc[c]); acc[c + 1] = fmaf(dd[e], w4.y, acc[c + 1]);
                acc[c + 2] = fmaf(dd[e], w4.z, acc[c + 2]); acc[c + 3] = fmaf(dd[e], w4.w, acc[c + 3]);
              }
            } else {
#pragma unroll
              for (int c = 0; c < CI; ++c) acc[c] = fmaf(dd[e], wp[c], acc[c]);
            }
          }
        }
      }
    }
    float* xp = dx + pos * CI;
    if constexpr (CI % 4 == 0) {
#pragma unroll
      for (int c = 0; c < CI; c += 4) *reinterpret_cast<float4*>(xp + c) = make_float4(acc[c], acc[c + 1], acc[c + 2], acc[c + 3]);
    } else {
#pragma unroll
      for (int c = 0; c < CI; ++c) xp[c] = acc[c];
    }
  }
}

// partials[chunk][co][ci][27]; block = ((ci, kh), chunk): 9 taps x all CO accumulators per thread, x row values loaded once
template <int CO>
__global__ __launch_bounds__(256) void conv2d_wgrad_row_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ partials, C2Geom g, int64_t pos_per_chunk) {
  __shared__ float red[4][CO * 9];
  const int ci = blockIdx.x / 3, kh = blockIdx.x % 3;
  const int64_t npos = (int64_t)g.B * g.Ho * g.Wo;
  const int64_t p0 = (int64_t)blockIdx.y * pos_per_chunk, p1 = min(npos, p0 + pos_per_chunk);
  float acc[CO][9];
#pragma unroll
  for (int c = 0; c < CO; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[c][k] = 0.f;
  for (int64_t pos = p0 + threadIdx.x; pos < p1; pos += 256) {
    const int ox = (int)(pos % g.Wo), oy = (int)((pos / g.Wo) % g.Ho), b = (int)(pos / ((int64_t)g.Wo * g.Ho));
    const int iy = oy * g.sh + kh - 1;
    if (iy < 0 || iy >= g.H) continue;
    float xv[9];
#pragma unroll
    for (int kw = 0; kw < 9; ++kw) {
      const int ix = ox * g.sw + kw - g.pw;
      xv[kw] = (ix >= 0 && ix < g.W) ? x[in_index(g, b, ci, iy, ix)] : 0.f;
    }
    const float4* dp = reinterpret_cast<const float4*>(dy + pos * CO);
#pragma unroll
    for (int c4 = 0; c4 < CO / 4; ++c4) {
      const float4 d = dp[c4];
      const float dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int kw = 0; kw < 9; ++kw) acc[c4 * 4 + e][kw] = fmaf(dd[e], xv[kw], acc[c4 * 4 + e][kw]);
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < CO; ++c)
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float s = wave_sum(acc[c][k]);
      if (lane == 0) red[wv][c * 9 + k] = s;
    }
  __syncthreads();
  for (int i = threadIdx.x; i < CO * 9; i += 256) {
    const int co = i / 9, kw = i % 9;
    partials[((int64_t)blockIdx.y * CO * g.Ci + (int64_t)co * g.Ci + ci) * 27 + kh * 9 + kw] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
  }
}

// dx (NHWC) [B][H][W][Ci]
__global__ __launch_bounds__(256) void conv2d_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                           float* __restrict__ dx, C2Geom g) {
  const int64_t total = (int64_t)g.B * g.H * g.W * g.Ci;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ci = (int)(i % g.Ci);
    const int64_t pos = i / g.Ci;
    const int ix = (int)(pos % g.W), iy = (int)((pos / g.W) % g.H), b = (int)(pos / ((int64_t)g.W * g.H));
    float acc = 0.f;
    for (int kh = 0; kh < 3; ++kh) {
      const int ty = iy + 1 - kh;
      if (ty < 0 || ty % g.sh != 0) continue;
      const int oy = ty / g.sh;
      if (oy >= g.Ho) continue;
#pragma unroll
      for (int kw = 0; kw < 9; ++kw) {
        const int tx = ix + g.pw - kw;
        if (tx < 0 || tx % g.sw != 0) continue;
        const int ox = tx / g.sw;
        if (ox >= g.Wo) continue;
        const float* dp = dy + (((int64_t)b * g.Ho + oy) * g.Wo + ox) * g.Co;
        for (int co = 0; co < g.Co; ++co) acc = fmaf(dp[co], w[(((int64_t)co * g.Ci + ci) * 3 + kh) * 9 + kw], acc);
      }
    }
    dx[i] = acc;
  }
}

// partials[chunk][co][ci][27]; block = (co*Ci+ci, chunk)
__global__ __launch_bounds__(256) void conv2d_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ partials, C2Geom g, int64_t pos_per_chunk) {
  __shared__ float red[4][27];
  const int pair = blockIdx.x, co = pair / g.Ci, ci = pair % g.Ci;
  const int64_t npos = (int64_t)g.B * g.Ho * g.Wo;
  const int64_t p0 = (int64_t)blockIdx.y * pos_per_chunk, p1 = min(npos, p0 + pos_per_chunk);
  float acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = 0.f;
  for (int64_t pos = p0 + threadIdx.x; pos < p1; pos += 256) {
    const int ox = (int)(pos % g.Wo), oy = (int)((pos / g.Wo) % g.Ho), b = (int)(pos / ((int64_t)g.Wo * g.Ho));
    const float d = dy[pos * g.Co + co];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int iy = oy * g.sh + kh - 1;
      const bool oky = iy >= 0 && iy < g.H;
#pragma unroll
      for (int kw = 0; kw < 9; ++kw) {
        const int ix = ox * g.sw + kw - g.pw;
        if (oky && ix >= 0 && ix < g.W) acc[kh * 9 + kw] = fmaf(d, x[in_index(g, b, ci, iy, ix)], acc[kh * 9 + kw]);
      }
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const float s = wave_sum(acc[k]);
    if (lane == 0) red[wv][k] = s;
  }
  __syncthreads();
  if (threadIdx.x < 27)
    partials[((int64_t)blockIdx.y * gridDim.x + pair) * 27 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// 16 outputs x 16 chunk phases per 256-thread block (one thread per output walking up to 256 chunks serially was a chain of dependent
// loads: 60 us for the 216 weights of the first layer); fixed summation order: deterministic
__global__ __launch_bounds__(256) void conv2d_wgrad_reduce_kernel(const float* __restrict__ partials, float* __restrict__ dw, int n, int nchunk, int beta) {
  __shared__ float red[16][17];
  const int o = threadIdx.x & 15, ph = threadIdx.x >> 4, i = blockIdx.x * 16 + o;
  float s = 0.f;
  if (i < n)
    for (int c = ph; c < nchunk; c += 16) s += partials[(int64_t)c * n + i];
  red[ph][o] = s;
  __syncthreads();
  if (ph == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < 16; ++p) t += red[p][o];
    dw[i] = beta ? dw[i] + t : t;
  }
}

static int make_c2geom(const char* who, C2Geom* g, int B, int Ci, int H, int W, int Co, int sh, int sw, int pw, int in_layout) {
  MAAVSS_CHECK_ARG(B > 0 && Ci > 0 && Co > 0 && H > 0 && W > 0, "%s: empty problem", who);
  MAAVSS_CHECK_ARG((sh == 1 || sh == 2) && (sw == 1 || sw == 2), "%s: stride must be 1 or 2", who);
  MAAVSS_CHECK_ARG(pw >= 0 && pw <= 8, "%s: bad padding", who);
  g->B = B; g->Ci = Ci; g->H = H; g->W = W; g->Co = Co; g->sh = sh; g->sw = sw; g->pw = pw; g->in_layout = in_layout;
  g->Ho = (H + 2 - 3) / sh + 1;
  g->Wo = (W + 2 * pw - 9) / sw + 1;
  MAAVSS_CHECK_ARG(g->Ho > 0 && g->Wo > 0, "%s: empty output", who);
  return MAAVSS_OK;
}

extern "C" int maavss_conv2d_fwd(const float* x, const float* w, float* y, int B, int Ci, int H, int W, int Co, int sh,
                                 int sw, int pw, int in_layout, void* stream) {
  MAAVSS_CHECK_ARG(x && w && y, "conv2d_fwd: null pointer");
  C2Geom g;
  if (int rc = make_c2geom("conv2d_fwd", &g, B, Ci, H, W, Co, sh, sw, pw, in_layout)) return rc;
  const int64_t total = (int64_t)B * g.Ho * g.Wo * Co, npos = (int64_t)B * g.Ho * g.Wo;
  const dim3 pgrid(min((int64_t)8192, (npos + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (Ci > 16) hipLaunchKernelGGL(conv2d_fwd_kernel, dim3(min((int64_t)4096, (total + 255) / 256)), dim3(256), 0, st, x, w, y, g);   // LDS weight image: C_in <= 16
  else if (Co == 4) hipLaunchKernelGGL(conv2d_fwd_pos_kernel<4>, pgrid, dim3(256), 0, st, x, w, y, g);
  else if (Co == 8) hipLaunchKernelGGL(conv2d_fwd_pos_kernel<8>, pgrid, dim3(256), 0, st, x, w, y, g);
  else if (Co == 16) hipLaunchKernelGGL(conv2d_fwd_pos_kernel<16>, pgrid, dim3(256), 0, st, x, w, y, g);
  else hipLaunchKernelGGL(conv2d_fwd_kernel, dim3(min((int64_t)4096, (total + 255) / 256)), dim3(256), 0, st, x, w, y, g);
  MAAVSS_LAUNCH_CHECK("conv2d_fwd_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_conv2d_dgrad(const float* dy, const float* w, float* dx, int B, int Ci, int H, int W, int Co, int sh,
                                   int sw, int pw, void* stream) {
  MAAVSS_CHECK_ARG(dy && w && dx, "conv2d_dgrad: null pointer");
  C2Geom g;
  if (int rc = make_c2geom("conv2d_dgrad", &g, B, Ci, H, W, Co, sh, sw, pw, 1)) return rc;
  const int64_t total = (int64_t)B * H * W * Ci, npos = (int64_t)B * H * W;
  const dim3 pgrid(min((int64_t)8192, (npos + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (Co > 16) hipLaunchKernelGGL(conv2d_dgrad_kernel, dim3(min((int64_t)4096, (total + 255) / 256)), dim3(256), 0, st, dy, w, dx, g);   // LDS weight image: C_out <= 16
  else if (Co % 4 == 0 && Ci == 2) hipLaunchKernelGGL(conv2d_dgrad_pos_kernel<2>, pgrid, dim3(256), 0, st, dy, w, dx, g);
  else if (Co % 4 == 0 && Ci == 4) hipLaunchKernelGGL(conv2d_dgrad_pos_kernel<4>, pgrid, dim3(256), 0, st, dy, w, dx, g);
  else if (Co % 4 == 0 && Ci == 8) hipLaunchKernelGGL(conv2d_dgrad_pos_kernel<8>, pgrid, dim3(256), 0, st, dy, w, dx, g);
  else if (Co % 4 == 0 && Ci == 16) hipLaunchKernelGGL(conv2d_dgrad_pos_kernel<16>, pgrid, dim3(256), 0, st, dy, w, dx, g);
  else hipLaunchKernelGGL(conv2d_dgrad_kernel, dim3(min((int64_t)4096, (total + 255) / 256)), dim3(256), 0, st, dy, w, dx, g);
  MAAVSS_LAUNCH_CHECK("conv2d_dgrad_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_conv2d_wgrad_nchunk(int B, int Ho, int Wo, int Ci, int Co) {
  const int64_t npos = (int64_t)B * Ho * Wo;
  // blocks = (ci, kh) rows x chunks for the row kernel (C_out in {4, 8, 16}), (co, ci) pairs x chunks otherwise
  int64_t n = (Co == 4 || Co == 8 || Co == 16) ? 1536 / ((int64_t)Ci * 3) : 1024 / ((int64_t)Ci * Co);
  if (n < 1) n = 1;
  if (n > (npos + 1023) / 1024) n = (npos + 1023) / 1024;
  if (n < 1) n = 1;
  return (int)n;
}

// ws: nchunk * Co*Ci*27 floats, nchunk = maavss_conv2d_wgrad_nchunk(...)
extern "C" int maavss_conv2d_wgrad(const float* x, const float* dy, float* dw, float* ws, int B, int Ci, int H, int W, int Co,
                                   int sh, int sw, int pw, int in_layout, int beta, void* stream) {
  MAAVSS_CHECK_ARG(x && dy && dw && ws, "conv2d_wgrad: null pointer");
  C2Geom g;
  if (int rc = make_c2geom("conv2d_wgrad", &g, B, Ci, H, W, Co, sh, sw, pw, in_layout)) return rc;
  const int nchunk = maavss_conv2d_wgrad_nchunk(B, g.Ho, g.Wo, Ci, Co);
  const int64_t npos = (int64_t)B * g.Ho * g.Wo;
  hipStream_t st = (hipStream_t)stream;
  const int64_t ppc = (npos + nchunk - 1) / nchunk;
  if (Co == 4) hipLaunchKernelGGL(conv2d_wgrad_row_kernel<4>, dim3(Ci * 3, nchunk), dim3(256), 0, st, x, dy, ws, g, ppc);
  else if (Co == 8) hipLaunchKernelGGL(conv2d_wgrad_row_kernel<8>, dim3(Ci * 3, nchunk), dim3(256), 0, st, x, dy, ws, g, ppc);
  else if (Co == 16) hipLaunchKernelGGL(conv2d_wgrad_row_kernel<16>, dim3(Ci * 3, nchunk), dim3(256), 0, st, x, dy, ws, g, ppc);
  else hipLaunchKernelGGL(conv2d_wgrad_kernel, dim3(Co * Ci, nchunk), dim3(256), 0, st, x, dy, ws, g, ppc);
  MAAVSS_LAUNCH_CHECK("conv2d_wgrad_kernel");
  const int n = Co * Ci * 27;
  hipLaunchKernelGGL(conv2d_wgrad_reduce_kernel, dim3(cdiv(n, 16)), dim3(256), 0, st, ws, dw, n, nchunk, beta);
  MAAVSS_LAUNCH_CHECK("conv2d_wgrad_reduce_kernel");
  return MAAVSS_OK;
}
