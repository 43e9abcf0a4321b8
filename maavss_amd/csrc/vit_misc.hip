// K1 (patch gather), K2 (LayerNorm) and K6 (attention-map post-process) of the ViT attention extractor.
//   vit_patchify      frames [F][3][H][W] f32 -> im2col rows [F*(n+1)][192] bf16 (row 0 of each frame = zeros for
//                     the CLS slot; the GEMM's periodic row table adds cls_token / conv bias + position embedding)
//   vit_layernorm     x [rows][384] f32 -> bf16, eps 1e-6, one wavefront per row (dino Block.norm1/norm2)
//   vit_attn_maps     reference video_attention.py:80-96 (reshape to [6,h,w], nearest x8 upsample, sum heads,
//                     times 1/max per frame) and av_dataset.py:328 (times 1/max per clip), fused:
//                     pass 1 per frame: head sum, frame max;  pass 2: clip max, upsampled store.
#include "common.h"

template <int MODE>
__global__ __launch_bounds__(256) void vit_patchify_kernel(const float* __restrict__ frames, bf16_t* __restrict__ a, int H,
                                                           int W, int hp, int wp, int64_t total) {
  // one thread = one (row, c, dy): 8 contiguous pixels -> 8 bf16 (16 B)
  const int ntok = hp * wp + 1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int seg = (int)(i % 24);  // c*8 + dy
    const int64_t row = i / 24;
    const int tok = (int)(row % ntok);
    const int64_t f = row / ntok;
    uint4 o = make_uint4(0, 0, 0, 0);
    if (tok > 0) {
      const int p = tok - 1, py = p / wp, px = p % wp, c = seg >> 3, dy = seg & 7;
      const float* src = frames + ((f * 3 + c) * H + py * 8 + dy) * (int64_t)W + px * 8;
      const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
      o.x = pack2<MODE>(v0.x, v0.y); o.y = pack2<MODE>(v0.z, v0.w); o.z = pack2<MODE>(v1.x, v1.y); o.w = pack2<MODE>(v1.z, v1.w);
    }
    *reinterpret_cast<uint4*>(a + row * 192 + seg * 8) = o;
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void vit_layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                            int64_t rows, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float2* xp = reinterpret_cast<const float2*>(x + row * 384);
  float2 v[3];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) { v[i] = xp[i * 64 + lane]; s += v[i].x + v[i].y; }
  const float mean = wave_sum(s) * (1.f / 384.f);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) { const float a = v[i].x - mean, b = v[i].y - mean; q += a * a + b * b; }
  const float rstd = rsqrtf(wave_sum(q) * (1.f / 384.f) + eps);
  unsigned* yp = reinterpret_cast<unsigned*>(y + row * 384);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = (i * 64 + lane) * 2;
    yp[i * 64 + lane] = pack2<MODE>((v[i].x - mean) * rstd * gamma[c] + beta[c], (v[i].y - mean) * rstd * gamma[c + 1] + beta[c + 1]);
  }
}

// pass 1: small[f][j] = (sum_h att[f][h][j]) * (1/max_j);  fmax[f] = max_j of the scaled map
// `nonfinite` (optional): sticky flag, set to 1 when a CLS-attention value is inf / NaN.  Every 16-bit overflow upstream (q, k, v,
// the GELU hidden as IEEE half > 65504) poisons the frame's residual stream and reaches this row as NaN, so this one check on
// 6 x n values per frame guards the whole extractor without touching its GEMM epilogues.
__global__ __launch_bounds__(256) void vit_maps_pass1_kernel(const float* __restrict__ att, float* __restrict__ small,
                                                             float* __restrict__ fmax, int heads, int n, int* __restrict__ nonfinite) {
  __shared__ float red[4];
  const int f = blockIdx.x, tid = threadIdx.x;
  const float* ap = att + (int64_t)f * heads * n;
  float mx = -1e30f;
  bool bad = false;
  for (int j = tid; j < n; j += 256) {
    float s = 0.f;
    for (int h = 0; h < heads; ++h) s += ap[h * n + j];
    small[(int64_t)f * n + j] = s;
    bad |= !(fabsf(s) <= 3.0e38f);      // false for inf and for NaN
    mx = fmaxf(mx, s);
  }
  if (nonfinite != nullptr && bad) atomicOr(nonfinite, 1);
  mx = wave_max(mx);
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float r = 1.f / mx;
  float m2 = -1e30f;
  for (int j = tid; j < n; j += 256) {
    const float v = small[(int64_t)f * n + j] * r;
    small[(int64_t)f * n + j] = v;
    m2 = fmaxf(m2, v);
  }
  __syncthreads();
  m2 = wave_max(m2);
  if ((tid & 63) == 0) red[tid >> 6] = m2;
  __syncthreads();
  if (tid == 0) fmax[f] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// attn_diff option (av_dataset.py:323-326): within each clip, frame t becomes frame t minus frame t-1 and frame 0
// becomes zero; the clip maximum (over the whole zero-padded canvas, hence >= 0) is stored in every fmax of the clip.
__global__ __launch_bounds__(256) void vit_maps_diff_kernel(float* __restrict__ small, float* __restrict__ fmax, int n, int T) {
  __shared__ float red[4];
  const int64_t c0 = (int64_t)blockIdx.x * T;
  float mx = 0.f;
  for (int j = threadIdx.x; j < n; j += 256) {
    for (int t = T - 1; t >= 1; --t) {
      const float d = small[(c0 + t) * n + j] - small[(c0 + t - 1) * n + j];
      small[(c0 + t) * n + j] = d;
      mx = fmaxf(mx, d);
    }
    small[c0 * n + j] = 0.f;
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  for (int t = threadIdx.x; t < T; t += 256) fmax[c0 + t] = mx;
}

// pass 2: out[f][0][y][x] = small[f][(y/8)*wp + x/8] * (1 / max over the clip's frames of fmax), zero outside the
// patch grid.  clip_frames = 0 -> no clip normalisation (VideoAttention._inference alone).
__global__ __launch_bounds__(256) void vit_maps_pass2_kernel(const float* __restrict__ small, const float* __restrict__ fmax,
                                                             float* __restrict__ out, int H, int W, int hp, int wp,
                                                             int clip_frames, int64_t total4) {
  const int W4 = W / 4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int x4 = (int)(i % W4), y = (int)((i / W4) % H);
    const int64_t f = i / ((int64_t)W4 * H);
    float sc = 1.f;
    if (clip_frames > 0) {
      const int64_t c0 = (f / clip_frames) * clip_frames;
      float m = -1e30f;
      for (int t = 0; t < clip_frames; ++t) m = fmaxf(m, fmax[c0 + t]);
      sc = 1.f / m;
    }
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    const int py = y >> 3;
    if (py < hp) {
      const float* sp = small + f * (int64_t)(hp * wp) + py * wp;
      const int x = x4 * 4;
      const int px = x >> 3;   // 4 consecutive pixels never straddle an 8-pixel patch
      if (px < wp) { const float v = sp[px] * sc; o = make_float4(v, v, v, v); }
    }
    reinterpret_cast<float4*>(out)[i] = o;
  }
}

extern "C" int maavss_vit_patchify(const float* frames, void* a, int64_t n_frames, int H, int W, int dtype, void* stream) {
  MAAVSS_CHECK_ARG(dtype == 0 || dtype == 2, "vit_patchify: dtype must be 0 (bf16) or 2 (f16)");
  MAAVSS_CHECK_ARG(frames && a && n_frames > 0, "vit_patchify: bad arguments");
  MAAVSS_CHECK_ARG(H >= 8 && W >= 8 && W % 4 == 0, "vit_patchify: frame must be at least 8x8 with W a multiple of 4");
  const int hp = H / 8, wp = W / 8;
  const int64_t total = n_frames * (hp * wp + 1) * 24;
  const dim3 grid((unsigned)((total + 255) / 256 > 65535 * 4 ? 65535 * 4 : (total + 255) / 256));
  if (dtype == 2) hipLaunchKernelGGL(vit_patchify_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, frames, (bf16_t*)a, H, W, hp, wp, total);
  else hipLaunchKernelGGL(vit_patchify_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, frames, (bf16_t*)a, H, W, hp, wp, total);
  MAAVSS_LAUNCH_CHECK("vit_patchify_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_vit_layernorm(const float* x, const float* gamma, const float* beta, void* y, int64_t rows, int dim,
                                    float eps, int dtype, void* stream) {
  MAAVSS_CHECK_ARG(x && gamma && beta && y && rows > 0, "vit_layernorm: bad arguments");
  MAAVSS_CHECK_ARG(dtype == 0 || dtype == 2, "vit_layernorm: dtype must be 0 (bf16) or 2 (f16)");
  MAAVSS_CHECK_ARG(dim == 384, "vit_layernorm: only dim 384 (ViT-S) is built");
  if (dtype == 2)
    hipLaunchKernelGGL(vit_layernorm_kernel<2>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, (bf16_t*)y, rows, eps);
  else
    hipLaunchKernelGGL(vit_layernorm_kernel<0>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, (bf16_t*)y, rows, eps);
  MAAVSS_LAUNCH_CHECK("vit_layernorm_kernel");
  return MAAVSS_OK;
}

// ws: n_frames * (hp*wp + 1) floats
extern "C" int maavss_vit_attn_maps_checked(const float* att, float* out, float* ws, int64_t n_frames, int heads, int H, int W,
                                            int clip_frames, int attn_diff, int32_t* nonfinite_flag, void* stream) {
  MAAVSS_CHECK_ARG(att && out && ws && n_frames > 0 && heads > 0, "vit_attn_maps: bad arguments");
  MAAVSS_CHECK_ARG(W % 4 == 0 && H >= 8 && W >= 8, "vit_attn_maps: W must be a multiple of 4");
  MAAVSS_CHECK_ARG(clip_frames == 0 || n_frames % clip_frames == 0, "vit_attn_maps: n_frames must be a multiple of clip_frames");
  MAAVSS_CHECK_ARG(!attn_diff || clip_frames > 0, "vit_attn_maps: attn_diff needs clip_frames > 0");
  const int hp = H / 8, wp = W / 8, n = hp * wp;
  float* small = ws;
  float* fmax = ws + n_frames * n;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(vit_maps_pass1_kernel, dim3((unsigned)n_frames), dim3(256), 0, st, att, small, fmax, heads, n, nonfinite_flag);
  MAAVSS_LAUNCH_CHECK("vit_maps_pass1_kernel");
  if (attn_diff) {
    hipLaunchKernelGGL(vit_maps_diff_kernel, dim3((unsigned)(n_frames / clip_frames)), dim3(256), 0, st, small, fmax, n, clip_frames);
    MAAVSS_LAUNCH_CHECK("vit_maps_diff_kernel");
  }
  const int64_t total4 = n_frames * H * (W / 4);
  hipLaunchKernelGGL(vit_maps_pass2_kernel, dim3((unsigned)((total4 + 255) / 256 > 16384 ? 16384 : (total4 + 255) / 256)),
                     dim3(256), 0, st, small, fmax, out, H, W, hp, wp, clip_frames, total4);
  MAAVSS_LAUNCH_CHECK("vit_maps_pass2_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_vit_attn_maps(const float* att, float* out, float* ws, int64_t n_frames, int heads, int H, int W,
                                    int clip_frames, int attn_diff, void* stream) {
  return maavss_vit_attn_maps_checked(att, out, ws, n_frames, heads, H, W, clip_frames, attn_diff, nullptr, stream);
}
