// K19: the biased 2-D convolutions of the phasegram variant avse_model.AV_Fusion_Model
// (avse_model.py:410-711) -- Conv2d / ConvTranspose2d with kernels (1,9) [phasegram encoder / decoder, :433,452] and
// (5,5) [STFT encoder / decoder, :494,591], strides in {1,2}^2, any padding, bias.
// One formulation serves all six operators.  A "small" map S [B][Hs][Ws][Cs] and a "big" map G [B][Hb][Wb][Cb] are
// tied by  by = sy*sh - ph + kh,  bx = sx*sw - pw + kw  and a weight w[cs][cb][kh][kw]:
//   gather_small:  S = bias + sum_{cb,kh,kw} G * w      = Conv2d forward (w = [Co][Ci])   = ConvTranspose2d input gradient
//   gather_big:    G = bias + sum_{cs,kh,kw} S * w      = ConvTranspose2d forward (w = [Ci][Co]) = Conv2d input gradient
//   wgrad:         dw[cs][cb][kh][kw] = sum_{b,sy,sx} S * G                                  (both operators)
// -- the PyTorch weight layouts of Conv2d ([Co][Ci]) and ConvTranspose2d ([Ci][Co]) are both [small][big].
// Tensors are addressed through explicit element strides (b, y, x, c), so the network's NCHW inputs / outputs and the
// channels-last activations (optionally channel-padded for the BatchNorm kernels) need no copies.
// A few MFLOP per clip: direct kernels, one thread per output element; latency / HBM-bound like conv2d.hip.
#include "common.h"

struct CGen {
  int B, Cs, Hs, Ws, Cb, Hb, Wb, kh, kw, sh, sw, ph, pw;
  int64_t ssb, ssy, ssx, ssc;   // strides of the small map
  int64_t gsb, gsy, gsx, gsc;   // strides of the big map
};

__global__ __launch_bounds__(256) void cgen_small_kernel(const float* __restrict__ G, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ S, CGen g) {
  const int64_t total = (int64_t)g.B * g.Hs * g.Ws * g.Cs;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cs = (int)(i % g.Cs);
    const int64_t pos = i / g.Cs;
    const int sx = (int)(pos % g.Ws), sy = (int)((pos / g.Ws) % g.Hs), b = (int)(pos / ((int64_t)g.Ws * g.Hs));
    float acc = bias ? bias[cs] : 0.f;
    for (int kh = 0; kh < g.kh; ++kh) {
      const int by = sy * g.sh - g.ph + kh;
      if (by < 0 || by >= g.Hb) continue;
      for (int kw = 0; kw < g.kw; ++kw) {
        const int bx = sx * g.sw - g.pw + kw;
        if (bx < 0 || bx >= g.Wb) continue;
        const float* gp = G + b * g.gsb + by * g.gsy + bx * g.gsx;
        const float* wp = w + ((int64_t)cs * g.Cb * g.kh + kh) * g.kw + kw;
        for (int cb = 0; cb < g.Cb; ++cb) acc = fmaf(gp[cb * g.gsc], wp[(int64_t)cb * g.kh * g.kw], acc);
      }
    }
    S[b * g.ssb + sy * g.ssy + sx * g.ssx + cs * g.ssc] = acc;
  }
}

__global__ __launch_bounds__(256) void cgen_big_kernel(const float* __restrict__ S, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ G, CGen g) {
  const int64_t total = (int64_t)g.B * g.Hb * g.Wb * g.Cb;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cb = (int)(i % g.Cb);
    const int64_t pos = i / g.Cb;
    const int bx = (int)(pos % g.Wb), by = (int)((pos / g.Wb) % g.Hb), b = (int)(pos / ((int64_t)g.Wb * g.Hb));
    float acc = bias ? bias[cb] : 0.f;
    for (int kh = 0; kh < g.kh; ++kh) {
      const int ty = by + g.ph - kh;
      if (ty < 0 || ty % g.sh != 0) continue;
      const int sy = ty / g.sh;
      if (sy >= g.Hs) continue;
      for (int kw = 0; kw < g.kw; ++kw) {
        const int tx = bx + g.pw - kw;
        if (tx < 0 || tx % g.sw != 0) continue;
        const int sx = tx / g.sw;
        if (sx >= g.Ws) continue;
        const float* sp = S + b * g.ssb + sy * g.ssy + sx * g.ssx;
        const float* wp = w + ((int64_t)cb * g.kh + kh) * g.kw + kw;
        for (int cs = 0; cs < g.Cs; ++cs) acc = fmaf(sp[cs * g.ssc], wp[(int64_t)cs * g.Cb * g.kh * g.kw], acc);
      }
    }
    G[b * g.gsb + by * g.gsy + bx * g.gsx + cb * g.gsc] = acc;
  }
}

// one block per (cs, cb) pair and chunk of small-map positions; partials[chunk][cs][cb][kh*kw]
#define CG_MAX_TAPS 25
__global__ __launch_bounds__(256) void cgen_wgrad_kernel(const float* __restrict__ S, const float* __restrict__ G,
                                                         float* __restrict__ partials, CGen g, int64_t pos_per_chunk) {
  __shared__ float red[4][CG_MAX_TAPS];
  const int pair = blockIdx.x, cs = pair / g.Cb, cb = pair % g.Cb;
  const int taps = g.kh * g.kw;
  const int64_t npos = (int64_t)g.B * g.Hs * g.Ws;
  const int64_t p0 = (int64_t)blockIdx.y * pos_per_chunk, p1 = min(npos, p0 + pos_per_chunk);
  float acc[CG_MAX_TAPS];
#pragma unroll
  for (int k = 0; k < CG_MAX_TAPS; ++k) acc[k] = 0.f;
  for (int64_t pos = p0 + threadIdx.x; pos < p1; pos += 256) {
    const int sx = (int)(pos % g.Ws), sy = (int)((pos / g.Ws) % g.Hs), b = (int)(pos / ((int64_t)g.Ws * g.Hs));
    const float sv = S[b * g.ssb + sy * g.ssy + sx * g.ssx + cs * g.ssc];
    const float* gb = G + b * g.gsb + cb * g.gsc;
#pragma unroll
    for (int k = 0; k < CG_MAX_TAPS; ++k) {
      if (k < taps) {
        const int kh = k / g.kw, kw = k % g.kw;
        const int by = sy * g.sh - g.ph + kh, bx = sx * g.sw - g.pw + kw;
        if (by >= 0 && by < g.Hb && bx >= 0 && bx < g.Wb) acc[k] = fmaf(sv, gb[by * g.gsy + bx * g.gsx], acc[k]);
      }
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < CG_MAX_TAPS; ++k) {
    const float s = wave_sum(acc[k]);
    if (lane == 0) red[wv][k] = s;
  }
  __syncthreads();
  if ((int)threadIdx.x < taps)
    partials[((int64_t)blockIdx.y * g.Cs * g.Cb + pair) * taps + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void cgen_reduce_kernel(const float* __restrict__ partials, float* __restrict__ out, int n,
                                                          int nchunk, int beta) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int c = 0; c < nchunk; ++c) s += partials[(int64_t)c * n + i];
  out[i] = beta ? out[i] + s : s;
}

// out[c] (+)= sum over rows of X[row * rs + c * cs]  -- bias gradients of the convolutions and of the Linear layers
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ X, float* __restrict__ out, int64_t rows, int C,
                                                          int64_t rs, int64_t cs, int beta) {
  __shared__ float red[4];
  const int c = blockIdx.x;
  float s = 0.f;
  for (int64_t r = threadIdx.x; r < rows; r += 256) s += X[r * rs + c * cs];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = red[0] + red[1] + red[2] + red[3];
    out[c] = beta ? out[c] + t : t;
  }
}

static int cg_fill(CGen& g, int B, int Cs, int Hs, int Ws, int Cb, int Hb, int Wb, int kh, int kw, int sh, int sw, int ph, int pw,
                   const int64_t* s_strides, const int64_t* g_strides, const char* who) {
  MAAVSS_CHECK_ARG(B > 0 && Cs > 0 && Hs > 0 && Ws > 0 && Cb > 0 && Hb > 0 && Wb > 0, "%s: empty tensor", who);
  MAAVSS_CHECK_ARG(kh >= 1 && kw >= 1 && kh * kw <= CG_MAX_TAPS, "%s: kernel %dx%d has more than %d taps", who, kh, kw, CG_MAX_TAPS);
  MAAVSS_CHECK_ARG(sh >= 1 && sw >= 1 && ph >= 0 && pw >= 0, "%s: bad stride / padding", who);
  MAAVSS_CHECK_ARG(s_strides && g_strides, "%s: stride arrays missing", who);
  // every small position must map inside the (padded) big map: (Hs-1)*sh - ph + kh - 1 <= Hb - 1 + ph  (conv arithmetic)
  MAAVSS_CHECK_ARG((Hs - 1) * sh + kh - 2 * ph <= Hb + (sh - 1) && (Ws - 1) * sw + kw - 2 * pw <= Wb + (sw - 1),
                   "%s: small map [%d,%d] does not fit big map [%d,%d]", who, Hs, Ws, Hb, Wb);
  g.B = B; g.Cs = Cs; g.Hs = Hs; g.Ws = Ws; g.Cb = Cb; g.Hb = Hb; g.Wb = Wb;
  g.kh = kh; g.kw = kw; g.sh = sh; g.sw = sw; g.ph = ph; g.pw = pw;
  g.ssb = s_strides[0]; g.ssy = s_strides[1]; g.ssx = s_strides[2]; g.ssc = s_strides[3];
  g.gsb = g_strides[0]; g.gsy = g_strides[1]; g.gsx = g_strides[2]; g.gsc = g_strides[3];
  return MAAVSS_OK;
}

static int cg_blocks(int64_t total) {
  const int64_t b = (total + 255) / 256;
  return (int)(b < 65535 * 8 ? b : 65535 * 8);
}

extern "C" int maavss_conv2d_gen_small(const float* big, const float* w, const float* bias, float* small, int B, int Cs, int Hs,
                                       int Ws, int Cb, int Hb, int Wb, int kh, int kw, int sh, int sw, int ph, int pw,
                                       const int64_t* small_strides, const int64_t* big_strides, void* stream) {
  MAAVSS_CHECK_ARG(big && w && small, "conv2d_gen_small: null pointer");
  CGen g;
  if (int rc = cg_fill(g, B, Cs, Hs, Ws, Cb, Hb, Wb, kh, kw, sh, sw, ph, pw, small_strides, big_strides, "conv2d_gen_small")) return rc;
  hipLaunchKernelGGL(cgen_small_kernel, dim3(cg_blocks((int64_t)B * Hs * Ws * Cs)), dim3(256), 0, (hipStream_t)stream, big, w, bias, small, g);
  MAAVSS_LAUNCH_CHECK("cgen_small_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_conv2d_gen_big(const float* small, const float* w, const float* bias, float* big, int B, int Cs, int Hs,
                                     int Ws, int Cb, int Hb, int Wb, int kh, int kw, int sh, int sw, int ph, int pw,
                                     const int64_t* small_strides, const int64_t* big_strides, void* stream) {
  MAAVSS_CHECK_ARG(small && w && big, "conv2d_gen_big: null pointer");
  CGen g;
  if (int rc = cg_fill(g, B, Cs, Hs, Ws, Cb, Hb, Wb, kh, kw, sh, sw, ph, pw, small_strides, big_strides, "conv2d_gen_big")) return rc;
  hipLaunchKernelGGL(cgen_big_kernel, dim3(cg_blocks((int64_t)B * Hb * Wb * Cb)), dim3(256), 0, (hipStream_t)stream, small, w, bias, big, g);
  MAAVSS_LAUNCH_CHECK("cgen_big_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_conv2d_gen_wgrad_nchunk(int B, int Hs, int Ws) {
  const int64_t n = ((int64_t)B * Hs * Ws + 4095) / 4096;
  return (int)(n < 1 ? 1 : (n > 128 ? 128 : n));
}

extern "C" int maavss_conv2d_gen_wgrad(const float* small, const float* big, float* dw, float* ws, int B, int Cs, int Hs, int Ws,
                                       int Cb, int Hb, int Wb, int kh, int kw, int sh, int sw, int ph, int pw,
                                       const int64_t* small_strides, const int64_t* big_strides, int beta, void* stream) {
  MAAVSS_CHECK_ARG(small && big && dw && ws, "conv2d_gen_wgrad: null pointer");
  CGen g;
  if (int rc = cg_fill(g, B, Cs, Hs, Ws, Cb, Hb, Wb, kh, kw, sh, sw, ph, pw, small_strides, big_strides, "conv2d_gen_wgrad")) return rc;
  const int nchunk = maavss_conv2d_gen_wgrad_nchunk(B, Hs, Ws);
  const int64_t npos = (int64_t)B * Hs * Ws;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(cgen_wgrad_kernel, dim3(Cs * Cb, nchunk), dim3(256), 0, st, small, big, ws, g, (npos + nchunk - 1) / nchunk);
  MAAVSS_LAUNCH_CHECK("cgen_wgrad_kernel");
  const int n = Cs * Cb * kh * kw;
  hipLaunchKernelGGL(cgen_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ws, dw, n, nchunk, beta);
  MAAVSS_LAUNCH_CHECK("cgen_reduce_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_channel_sum(const float* x, float* out, int64_t rows, int C, int64_t row_stride, int64_t chan_stride, int beta,
                                  void* stream) {
  MAAVSS_CHECK_ARG(x && out && rows > 0 && C > 0, "channel_sum: bad arguments");
  hipLaunchKernelGGL(channel_sum_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, out, rows, C, row_stride, chan_stride, beta);
  MAAVSS_LAUNCH_CHECK("channel_sum_kernel");
  return MAAVSS_OK;
}
