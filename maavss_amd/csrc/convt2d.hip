// K11: the STFT decoder's ConvTranspose2d(k=(3,kw), kw in {9,10}, stride (sh,sw) in {1,2}^2, padding (1,4),
// output_padding (oph,opw), bias=False) layers (reference avse_model_final.py:155-193; used by audio_ae_forward
// :254-256 through stft_autoencoder).  16 -> 8 -> 4 -> 2 (-> 2) channels on at most [128 x 257] maps: a few MFLOP per
// clip, so -- like the encoder's Conv2d (conv2d.hip) -- these are direct kernels, one thread per output element,
// operands through L1/L2; HBM/latency-bound.
//   y[b,oy,ox,co] = sum_{ci,kh,kw} x[b,iy,ix,ci] * W[ci,co,kh,kw],   oy = iy*sh - 1 + kh,  ox = ix*sw - 4 + kw
// Activations are NHWC; the LAST decoder layer writes (and takes its output gradient in) the network's NCHW layout
// (out_layout 0).  Weights stay in the reference layout [Ci][Co][3][kw].
#include "common.h"

struct CT2Geom {
  int B, Ci, Hi, Wi, Co, Ho, Wo, kw, sh, sw, out_layout;
};

__device__ __forceinline__ int64_t ct_out_index(const CT2Geom& g, int b, int co, int oy, int ox) {
  return g.out_layout ? (((int64_t)b * g.Ho + oy) * g.Wo + ox) * g.Co + co : (((int64_t)b * g.Co + co) * g.Ho + oy) * g.Wo + ox;
}

__global__ __launch_bounds__(256) void convt2d_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          float* __restrict__ y, CT2Geom g) {
  const int64_t total = (int64_t)g.B * g.Ho * g.Wo * g.Co;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    // decompose in NHWC order (co fastest) whatever the storage layout: neighbouring threads share input pixels
    const int co = (int)(i % g.Co);
    const int64_t pos = i / g.Co;
    const int ox = (int)(pos % g.Wo), oy = (int)((pos / g.Wo) % g.Ho), b = (int)(pos / ((int64_t)g.Wo * g.Ho));
    float acc = 0.f;
    for (int kh = 0; kh < 3; ++kh) {
      const int ty = oy + 1 - kh;
      if (ty < 0 || ty % g.sh != 0) continue;
      const int iy = ty / g.sh;
      if (iy >= g.Hi) continue;
      for (int kx = 0; kx < g.kw; ++kx) {
        const int tx = ox + 4 - kx;
        if (tx < 0 || tx % g.sw != 0) continue;
        const int ix = tx / g.sw;
        if (ix >= g.Wi) continue;
        const float* xp = x + (((int64_t)b * g.Hi + iy) * g.Wi + ix) * g.Ci;
        for (int ci = 0; ci < g.Ci; ++ci) acc = fmaf(xp[ci], w[(((int64_t)ci * g.Co + co) * 3 + kh) * g.kw + kx], acc);
      }
    }
    y[ct_out_index(g, b, co, oy, ox)] = acc;
  }
}

// dx (NHWC) [B][Hi][Wi][Ci] = sum_{co,kh,kw} dy[b, iy*sh-1+kh, ix*sw-4+kw, co] * W[ci,co,kh,kw]
__global__ __launch_bounds__(256) void convt2d_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                            float* __restrict__ dx, CT2Geom g) {
  const int64_t total = (int64_t)g.B * g.Hi * g.Wi * g.Ci;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ci = (int)(i % g.Ci);
    const int64_t pos = i / g.Ci;
    const int ix = (int)(pos % g.Wi), iy = (int)((pos / g.Wi) % g.Hi), b = (int)(pos / ((int64_t)g.Wi * g.Hi));
    float acc = 0.f;
    for (int kh = 0; kh < 3; ++kh) {
      const int oy = iy * g.sh - 1 + kh;
      if (oy < 0 || oy >= g.Ho) continue;
      for (int kx = 0; kx < g.kw; ++kx) {
        const int ox = ix * g.sw - 4 + kx;
        if (ox < 0 || ox >= g.Wo) continue;
        for (int co = 0; co < g.Co; ++co)
          acc = fmaf(dy[ct_out_index(g, b, co, oy, ox)], w[(((int64_t)ci * g.Co + co) * 3 + kh) * g.kw + kx], acc);
      }
    }
    dx[i] = acc;
  }
}

// dW[ci][co][kh][kx] = sum_{b,iy,ix} x[b,iy,ix,ci] * dy[b, iy*sh-1+kh, ix*sw-4+kx, co].
// One block per (ci, co) pair and position chunk; partials[chunk][ci][co][3*kw], reduced by convt2d_wgrad_reduce_kernel.
#define CT_MAX_TAPS 30
__global__ __launch_bounds__(256) void convt2d_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ partials, CT2Geom g, int64_t pos_per_chunk) {
  __shared__ float red[4][CT_MAX_TAPS];
  const int pair = blockIdx.x, ci = pair / g.Co, co = pair % g.Co;
  const int taps = 3 * g.kw;
  const int64_t npos = (int64_t)g.B * g.Hi * g.Wi;
  const int64_t p0 = (int64_t)blockIdx.y * pos_per_chunk, p1 = min(npos, p0 + pos_per_chunk);
  float acc[CT_MAX_TAPS];
#pragma unroll
  for (int k = 0; k < CT_MAX_TAPS; ++k) acc[k] = 0.f;
  for (int64_t pos = p0 + threadIdx.x; pos < p1; pos += 256) {
    const int ix = (int)(pos % g.Wi), iy = (int)((pos / g.Wi) % g.Hi), b = (int)(pos / ((int64_t)g.Wi * g.Hi));
    const float xv = x[pos * g.Ci + ci];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int oy = iy * g.sh - 1 + kh;
      const bool yok = oy >= 0 && oy < g.Ho;
#pragma unroll
      for (int kx = 0; kx < 10; ++kx) {
        const int ox = ix * g.sw - 4 + kx;
        if (yok && kx < g.kw && ox >= 0 && ox < g.Wo) acc[kh * 10 + kx] = fmaf(xv, dy[ct_out_index(g, b, co, oy, ox)], acc[kh * 10 + kx]);
      }
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < CT_MAX_TAPS; ++k) {
    const float s = wave_sum(acc[k]);
    if (lane == 0) red[wv][k] = s;
  }
  __syncthreads();
  if ((int)threadIdx.x < taps) {
    const int kh = threadIdx.x / g.kw, kx = threadIdx.x % g.kw;
    const int k = kh * 10 + kx;
    partials[((int64_t)blockIdx.y * g.Ci * g.Co + pair) * taps + threadIdx.x] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
  }
}

__global__ __launch_bounds__(256) void convt2d_wgrad_reduce_kernel(const float* __restrict__ partials, float* __restrict__ dw,
                                                                   int n, int nchunk, int beta) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int c = 0; c < nchunk; ++c) s += partials[(int64_t)c * n + i];
  dw[i] = beta ? dw[i] + s : s;
}

static int ct_geom(CT2Geom& g, int B, int Ci, int Hi, int Wi, int Co, int kw, int sh, int sw, int oph, int opw, int out_layout,
                   const char* who) {
  MAAVSS_CHECK_ARG(B > 0 && Ci > 0 && Hi > 0 && Wi > 0 && Co > 0, "%s: empty tensor", who);
  MAAVSS_CHECK_ARG(kw == 9 || kw == 10, "%s: kernel width must be 9 or 10 (got %d)", who, kw);
  MAAVSS_CHECK_ARG((sh == 1 || sh == 2) && (sw == 1 || sw == 2), "%s: strides must be 1 or 2", who);
  MAAVSS_CHECK_ARG(oph >= 0 && oph < sh + (sh == 1) && opw >= 0 && opw < sw + (sw == 1) && oph <= 1 && opw <= 1, "%s: bad output_padding", who);
  MAAVSS_CHECK_ARG(out_layout == 0 || out_layout == 1, "%s: out_layout must be 0 (NCHW) or 1 (NHWC)", who);
  g.B = B; g.Ci = Ci; g.Hi = Hi; g.Wi = Wi; g.Co = Co; g.kw = kw; g.sh = sh; g.sw = sw; g.out_layout = out_layout;
  g.Ho = (Hi - 1) * sh - 2 + 3 + oph;
  g.Wo = (Wi - 1) * sw - 8 + kw + opw;
  MAAVSS_CHECK_ARG(g.Ho > 0 && g.Wo > 0, "%s: output size is zero", who);
  return MAAVSS_OK;
}

static int ct_blocks(int64_t total) {
  const int64_t b = (total + 255) / 256;
  return (int)(b < 65535 * 8 ? b : 65535 * 8);
}

extern "C" int maavss_convt2d_out_size(int Hi, int Wi, int kw, int sh, int sw, int oph, int opw, int* Ho, int* Wo) {
  MAAVSS_CHECK_ARG(Ho && Wo, "convt2d_out_size: null pointer");
  *Ho = (Hi - 1) * sh - 2 + 3 + oph;
  *Wo = (Wi - 1) * sw - 8 + kw + opw;
  return MAAVSS_OK;
}

extern "C" int maavss_convt2d_fwd(const float* x, const float* w, float* y, int B, int Ci, int Hi, int Wi, int Co, int kw,
                                  int sh, int sw, int oph, int opw, int out_layout, void* stream) {
  MAAVSS_CHECK_ARG(x && w && y, "convt2d_fwd: null pointer");
  CT2Geom g;
  if (int rc = ct_geom(g, B, Ci, Hi, Wi, Co, kw, sh, sw, oph, opw, out_layout, "convt2d_fwd")) return rc;
  const int64_t total = (int64_t)B * g.Ho * g.Wo * Co;
  hipLaunchKernelGGL(convt2d_fwd_kernel, dim3(ct_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, w, y, g);
  MAAVSS_LAUNCH_CHECK("convt2d_fwd_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_convt2d_dgrad(const float* dy, const float* w, float* dx, int B, int Ci, int Hi, int Wi, int Co, int kw,
                                    int sh, int sw, int oph, int opw, int out_layout, void* stream) {
  MAAVSS_CHECK_ARG(dy && w && dx, "convt2d_dgrad: null pointer");
  CT2Geom g;
  if (int rc = ct_geom(g, B, Ci, Hi, Wi, Co, kw, sh, sw, oph, opw, out_layout, "convt2d_dgrad")) return rc;
  const int64_t total = (int64_t)B * Hi * Wi * Ci;
  hipLaunchKernelGGL(convt2d_dgrad_kernel, dim3(ct_blocks(total)), dim3(256), 0, (hipStream_t)stream, dy, w, dx, g);
  MAAVSS_LAUNCH_CHECK("convt2d_dgrad_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_convt2d_wgrad_nchunk(int B, int Hi, int Wi) {
  const int64_t npos = (int64_t)B * Hi * Wi;
  const int64_t n = (npos + 8191) / 8192;
  return (int)(n < 1 ? 1 : (n > 256 ? 256 : n));
}

extern "C" int maavss_convt2d_wgrad(const float* x, const float* dy, float* dw, float* ws, int B, int Ci, int Hi, int Wi, int Co,
                                    int kw, int sh, int sw, int oph, int opw, int out_layout, int beta, void* stream) {
  MAAVSS_CHECK_ARG(x && dy && dw && ws, "convt2d_wgrad: null pointer");
  CT2Geom g;
  if (int rc = ct_geom(g, B, Ci, Hi, Wi, Co, kw, sh, sw, oph, opw, out_layout, "convt2d_wgrad")) return rc;
  const int nchunk = maavss_convt2d_wgrad_nchunk(B, Hi, Wi);
  const int64_t npos = (int64_t)B * Hi * Wi;
  const int64_t per = (npos + nchunk - 1) / nchunk;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(convt2d_wgrad_kernel, dim3(Ci * Co, nchunk), dim3(256), 0, st, x, dy, ws, g, per);
  MAAVSS_LAUNCH_CHECK("convt2d_wgrad_kernel");
  const int n = Ci * Co * 3 * kw;
  hipLaunchKernelGGL(convt2d_wgrad_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ws, dw, n, nchunk, beta);
  MAAVSS_LAUNCH_CHECK("convt2d_wgrad_reduce_kernel");
  return MAAVSS_OK;
}
