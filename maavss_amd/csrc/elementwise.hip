// K15 / K16 and small elementwise helpers (all HBM-bound, float4-vectorised grid-stride kernels):
//   act_bwd      dz = dout * act'(out) for the tanh / sigmoid heads (reference avse_model_final.py:246-249,264-268)
//   mse_pair     the two MSE terms of train_avse_frames.py:166-170 and their gradients in one pass
//   adam_step    torch.optim.Adam (train_avse_frames.py:92,180; betas .9/.999, eps 1e-8, no weight decay) over
//                one flat f32 parameter buffer (multi-tensor by construction: all parameters live in one buffer)
#include "common.h"

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                      float* __restrict__ dz, int64_t n, int act) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float o = out[i];
    dz[i] = dout[i] * (act == 1 ? 1.f - o * o : o * (1.f - o));
  }
}

// partials[blk] = sum (pred-target)^2 over the block's range; dpred = gscale * (pred - target) (nullable)
__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                          float* __restrict__ dpred, float gscale, int64_t n,
                                                          float* __restrict__ partials) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = pred[i] - target[i];
    s += d * d;
    if (dpred != nullptr) dpred[i] = gscale * d;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// losses[0] = a_loss, [1] = v_loss, [2] = (a_loss + coeff * v_loss) / num_seq
__global__ void mse_finalize_kernel(const float* __restrict__ pa, int na_blk, double na, const float* __restrict__ pv,
                                    int nv_blk, double nv, float coeff, float inv_num_seq, float* __restrict__ losses) {
  if (threadIdx.x != 0) return;
  double sa = 0.0, sv = 0.0;
  for (int i = 0; i < na_blk; ++i) sa += (double)pa[i];
  for (int i = 0; i < nv_blk; ++i) sv += (double)pv[i];
  const double la = sa / na, lv = sv / nv;
  losses[0] = (float)la;
  losses[1] = (float)lv;
  losses[2] = (float)((la + (double)coeff * lv) * (double)inv_num_seq);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n4, int64_t n, float lr_bc1, float b1,
                                                   float b2, float eps, float inv_sqrt_bc2, float gscale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
#define ADAM1(f)                                                       \
  {                                                                    \
    const float gr = gg.f * gscale;                                    \
    mm.f = b1 * mm.f + (1.f - b1) * gr;                                \
    vv.f = b2 * vv.f + (1.f - b2) * gr * gr;                           \
    pp.f -= lr_bc1 * mm.f / (sqrtf(vv.f) * inv_sqrt_bc2 + eps);        \
  }
    ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (int64_t i = n4 * 4; i < n; ++i) {
      const float gr = g[i] * gscale;
      m[i] = b1 * m[i] + (1.f - b1) * gr;
      v[i] = b2 * v[i] + (1.f - b2) * gr * gr;
      p[i] -= lr_bc1 * m[i] / (sqrtf(v[i]) * inv_sqrt_bc2 + eps);
    }
  }
}

// EXTENSION (not in the reference): AdaptiveAvgPool2d that ends the STFT encoder when the frame size is
// not constructible by halving (224^2, 384^2).  x NHWC [B][H][W][C]; out / dout addressed as
// b*os_b + (oy*Wo+ox)*os_p + c*os_c (so it can write the LSTM sequence buffer directly).
__global__ __launch_bounds__(256) void adaptive_pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, int B,
                                                                int H, int W, int C, int Ho, int Wo, int64_t osb,
                                                                int64_t osp, int64_t osc) {
  const int64_t total = (int64_t)B * Ho * Wo * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C), ox = (int)((i / C) % Wo), oy = (int)((i / ((int64_t)C * Wo)) % Ho), b = (int)(i / ((int64_t)C * Wo * Ho));
    const int y0 = (oy * H) / Ho, y1 = ((oy + 1) * H + Ho - 1) / Ho, x0 = (ox * W) / Wo, x1 = ((ox + 1) * W + Wo - 1) / Wo;
    float s = 0.f;
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) s += x[(((int64_t)b * H + yy) * W + xx) * C + c];
    out[b * osb + ((int64_t)oy * Wo + ox) * osp + c * osc] = s / (float)((y1 - y0) * (x1 - x0));
  }
}
__global__ __launch_bounds__(256) void adaptive_pool_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, int B,
                                                                int H, int W, int C, int Ho, int Wo, int64_t osb,
                                                                int64_t osp, int64_t osc) {
  const int64_t total = (int64_t)B * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C), ix = (int)((i / C) % W), iy = (int)((i / ((int64_t)C * W)) % H), b = (int)(i / ((int64_t)C * W * H));
    float s = 0.f;
    for (int oy = 0; oy < Ho; ++oy) {
      const int y0 = (oy * H) / Ho, y1 = ((oy + 1) * H + Ho - 1) / Ho;
      if (iy < y0 || iy >= y1) continue;
      for (int ox = 0; ox < Wo; ++ox) {
        const int x0 = (ox * W) / Wo, x1 = ((ox + 1) * W + Wo - 1) / Wo;
        if (ix < x0 || ix >= x1) continue;
        s += dout[b * osb + ((int64_t)oy * Wo + ox) * osp + c * osc] / (float)((y1 - y0) * (x1 - x0));
      }
    }
    dx[i] = s;
  }
}

static inline int ew_grid(int64_t n) { return (int)(n / 256 + 1 > 4096 ? 4096 : n / 256 + 1); }

extern "C" int maavss_act_bwd(const float* dout, const float* out, float* dz, int64_t n, int act, void* stream) {
  MAAVSS_CHECK_ARG(dout && out && dz && n > 0, "act_bwd: bad arguments");
  MAAVSS_CHECK_ARG(act == 1 || act == 2, "act_bwd: act must be 1 (tanh) or 2 (sigmoid)");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, dout, out, dz, n, act);
  MAAVSS_LAUNCH_CHECK("act_bwd_kernel");
  return MAAVSS_OK;
}

#define MSE_BLK 512
// ws: 2*MSE_BLK floats.  d_a / d_v nullable.  Gradients are those of losses[2].
extern "C" int maavss_mse_pair(const float* a_pred, const float* a_tgt, int64_t na, const float* v_pred, const float* v_tgt,
                               int64_t nv, float coeff, float inv_num_seq, float* d_a, float* d_v, float* losses, float* ws,
                               void* stream) {
  MAAVSS_CHECK_ARG(a_pred && a_tgt && v_pred && v_tgt && losses && ws && na > 0 && nv > 0, "mse_pair: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int ga = (int)((na + 4095) / 4096 > MSE_BLK ? MSE_BLK : (na + 4095) / 4096);
  const int gv = (int)((nv + 4095) / 4096 > MSE_BLK ? MSE_BLK : (nv + 4095) / 4096);
  hipLaunchKernelGGL(mse_partial_kernel, dim3(ga), dim3(256), 0, st, a_pred, a_tgt, d_a, 2.f * inv_num_seq / (float)na, na, ws);
  hipLaunchKernelGGL(mse_partial_kernel, dim3(gv), dim3(256), 0, st, v_pred, v_tgt, d_v, 2.f * coeff * inv_num_seq / (float)nv, nv, ws + MSE_BLK);
  hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(64), 0, st, ws, ga, (double)na, ws + MSE_BLK, gv, (double)nv, coeff, inv_num_seq, losses);
  MAAVSS_LAUNCH_CHECK("mse_pair");
  return MAAVSS_OK;
}

extern "C" int maavss_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                float eps, int64_t step, float grad_scale, void* stream) {
  MAAVSS_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "adam_step: bad arguments");
  MAAVSS_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_step: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n / 4, n, (float)(lr / bc1),
                     beta1, beta2, eps, (float)(1.0 / sqrt(bc2)), grad_scale);
  MAAVSS_LAUNCH_CHECK("adam_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_adaptive_pool_fwd(const float* x, float* out, int B, int H, int W, int C, int Ho, int Wo, int64_t os_b,
                                        int64_t os_p, int64_t os_c, void* stream) {
  MAAVSS_CHECK_ARG(x && out && B > 0 && H >= Ho && W >= Wo && Ho > 0 && Wo > 0 && C > 0, "adaptive_pool_fwd: bad arguments");
  hipLaunchKernelGGL(adaptive_pool_fwd_kernel, dim3(ew_grid((int64_t)B * Ho * Wo * C)), dim3(256), 0, (hipStream_t)stream, x, out,
                     B, H, W, C, Ho, Wo, os_b, os_p, os_c);
  MAAVSS_LAUNCH_CHECK("adaptive_pool_fwd_kernel");
  return MAAVSS_OK;
}
extern "C" int maavss_adaptive_pool_bwd(const float* dout, float* dx, int B, int H, int W, int C, int Ho, int Wo, int64_t os_b,
                                        int64_t os_p, int64_t os_c, void* stream) {
  MAAVSS_CHECK_ARG(dout && dx && B > 0 && H >= Ho && W >= Wo && Ho > 0 && Wo > 0 && C > 0, "adaptive_pool_bwd: bad arguments");
  hipLaunchKernelGGL(adaptive_pool_bwd_kernel, dim3(ew_grid((int64_t)B * H * W * C)), dim3(256), 0, (hipStream_t)stream, dout, dx,
                     B, H, W, C, Ho, Wo, os_b, os_p, os_c);
  MAAVSS_LAUNCH_CHECK("adaptive_pool_bwd_kernel");
  return MAAVSS_OK;
}

// ---- Linear bias + LeakyReLU(slope) of the phasegram variant (avse_model.py:660-663, 613-621): z = act(z + bias) in
// place over [rows][n]; act 0 = identity, 3 = LeakyReLU(slope).  leaky_bwd: dz = dout * (out > 0 ? 1 : slope).
__global__ __launch_bounds__(256) void bias_act_kernel(float* __restrict__ z, const float* __restrict__ bias, int64_t total, int n,
                                                       int act, float slope) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    float v = z[i] + (bias ? bias[i % n] : 0.f);
    if (act == 3) v = v > 0.f ? v : v * slope;
    z[i] = v;
  }
}
__global__ __launch_bounds__(256) void leaky_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                        float* __restrict__ dz, int64_t total, float slope) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
    dz[i] = out[i] > 0.f ? dout[i] : dout[i] * slope;
}

extern "C" int maavss_bias_act_fwd(float* z, const float* bias, int64_t rows, int n, int act, float slope, void* stream) {
  MAAVSS_CHECK_ARG(z && rows > 0 && n > 0 && (act == 0 || act == 3), "bias_act_fwd: bad arguments (act must be 0 or 3)");
  hipLaunchKernelGGL(bias_act_kernel, dim3(ew_grid(rows * n)), dim3(256), 0, (hipStream_t)stream, z, bias, rows * n, n, act, slope);
  MAAVSS_LAUNCH_CHECK("bias_act_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_leaky_bwd(const float* dout, const float* out, float* dz, int64_t n, float slope, void* stream) {
  MAAVSS_CHECK_ARG(dout && out && dz && n > 0 && slope > 0.f, "leaky_bwd: bad arguments (slope must be > 0: the sign of out is the sign of z)");
  hipLaunchKernelGGL(leaky_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, dout, out, dz, n, slope);
  MAAVSS_LAUNCH_CHECK("leaky_bwd_kernel");
  return MAAVSS_OK;
}

// ---- bf16 wire format of the gradient all-reduce (K18, optional; SURVEY.md 5: "bf16 gradient compression"): the flat f32 gradient
// bucket is rounded to bf16 for the collective and widened back into the f32 master buffer afterwards.  8 elements per thread.
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
    reinterpret_cast<uint4*>(dst)[i] = make_uint4(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w), pack_bf2(b.x, b.y), pack_bf2(b.z, b.w));
  }
}
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const uint4 u = reinterpret_cast<const uint4*>(src)[i];
    reinterpret_cast<float4*>(dst)[2 * i] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                                                        __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    reinterpret_cast<float4*>(dst)[2 * i + 1] = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u),
                                                            __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u));
  }
}
extern "C" int maavss_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
  MAAVSS_CHECK_ARG(src && dst && n > 0 && n % 8 == 0, "f32_to_bf16: n must be a positive multiple of 8");
  MAAVSS_CHECK_ARG(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "f32_to_bf16: 16-byte aligned buffers");
  const int64_t n8 = n / 8;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((unsigned)(n8 / 256 + 1 > 4096 ? 4096 : n8 / 256 + 1)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n8);
  MAAVSS_LAUNCH_CHECK("f32_to_bf16_kernel");
  return MAAVSS_OK;
}
extern "C" int maavss_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream) {
  MAAVSS_CHECK_ARG(src && dst && n > 0 && n % 8 == 0, "bf16_to_f32: n must be a positive multiple of 8");
  MAAVSS_CHECK_ARG(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "bf16_to_f32: 16-byte aligned buffers");
  const int64_t n8 = n / 8;
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)(n8 / 256 + 1 > 4096 ? 4096 : n8 / 256 + 1)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, n8);
  MAAVSS_LAUNCH_CHECK("bf16_to_f32_kernel");
  return MAAVSS_OK;
}
