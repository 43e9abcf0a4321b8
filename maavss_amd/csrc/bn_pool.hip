// K9/K10 (normalisation part): train-mode BatchNorm + MaxPool(1,p,p) + LeakyReLU(0.01)  (visual encoder,
// reference avse_model_final.py:35-37 ... 55-57: pool BEFORE the activation) and BatchNorm2d + Tanh (STFT
// encoder, :103-104; p = 1).  All tensors channels-last f32; HBM-bound elementwise/reduction kernels.
//
//   bn_stats          per-channel (sum, sum^2) partials of a [M][C] tensor (conv kernels can also emit them)
//   bn_finalize       partials -> mean / invstd (biased var, eps) + running stats update (momentum, unbiased var)
//   bn_pool_act_fwd   out = act(maxpool_p(gamma * (y - mean) * invstd + beta)), argmax index per window
//   bn_pool_act_bwd_reduce   g = dout * act'(out) routed to the argmax; per-channel partials of (sum g, sum g*xhat)
//   bn_bwd_finalize   partials -> dgamma, dbeta, and the three per-channel constants of the dx formula
//   bn_pool_act_bwd_dx       dy = gamma*invstd * (g_at_argmax - mean(g) - xhat * mean(g*xhat))   (dense, full res)
#include "common.h"

#define ACT_LEAKY 0
#define BN_INV_MIN_GAMMA 1e-2f
#define ACT_TANH 1

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ y, float* __restrict__ partials,
                                                       int64_t M, int C, int rows_per_block) {
  // thread -> 4 channels (c4 = tid % (C/4)), row phase tid / (C/4); C is a power of two in [4, 64]
  __shared__ float4 red[2][256];
  const int tid = threadIdx.x, C4 = C >> 2, c4 = tid % C4, ph = tid / C4, nph = 256 / C4;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  for (int64_t r = r0 + ph; r < r1; r += nph) {
    const float4 v = *reinterpret_cast<const float4*>(y + r * C + c4 * 4);
    s1.x += v.x; s1.y += v.y; s1.z += v.z; s1.w += v.w;
    s2.x += v.x * v.x; s2.y += v.y * v.y; s2.z += v.z * v.z; s2.w += v.w * v.w;
  }
  red[0][tid] = s1;
  red[1][tid] = s2;
  __syncthreads();
  if (tid < C4) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    for (int p = 0; p < nph; ++p) {
      const float4 u = red[0][p * C4 + tid], w = red[1][p * C4 + tid];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
      b.x += w.x; b.y += w.y; b.z += w.z; b.w += w.w;
    }
    *reinterpret_cast<float4*>(partials + (int64_t)blockIdx.x * 2 * C + tid * 4) = a;
    *reinterpret_cast<float4*>(partials + (int64_t)blockIdx.x * 2 * C + C + tid * 4) = b;
  }
}

// level-1 reduction of many partial rows: block b sums rows [b*rpb, (b+1)*rpb) of [nblk][2C] into out[b][2C]
__global__ __launch_bounds__(256) void bn_reduce_rows_kernel(const float* __restrict__ partials, float* __restrict__ out,
                                                             int nblk, int C2, int rpb) {
  __shared__ double red[256];
  const int tid = threadIdx.x, idx = tid % C2, ph = tid / C2, nph = 256 / C2;
  const int r0 = blockIdx.x * rpb, r1 = min(nblk, r0 + rpb);
  double s = 0.0;
  for (int r = r0 + ph; r < r1; r += nph) s += (double)partials[(int64_t)r * C2 + idx];
  red[tid] = s;
  __syncthreads();
  if (tid < C2) {
    double a = 0.0;
    for (int p = 0; p < nph; ++p) a += red[p * C2 + tid];
    out[(int64_t)blockIdx.x * C2 + tid] = (float)a;
  }
}

// Per-channel sums of the nblk partial rows [2][C] in double: one 1024-thread block, 1024 / C row phases per channel
// (a single thread per channel walking the rows took 45-50 us per BatchNorm layer, 0.85 ms per step in all).
__device__ __forceinline__ void bn_sum_partials(const float* __restrict__ partials, int nblk, int C, double& s1, double& s2) {
  __shared__ double red[2][1024];
  const int tid = threadIdx.x, c = tid % C, ph = tid / C, nph = 1024 / C;
  double a1 = 0.0, a2 = 0.0;
  for (int b = ph; b < nblk; b += nph) {
    a1 += (double)partials[(int64_t)b * 2 * C + c];
    a2 += (double)partials[(int64_t)b * 2 * C + C + c];
  }
  red[0][tid] = a1;
  red[1][tid] = a2;
  __syncthreads();
  s1 = 0.0;
  s2 = 0.0;
  if (tid < C)
    for (int p = 0; p < nph; ++p) {
      s1 += red[0][p * C + tid];
      s2 += red[1][p * C + tid];
    }
}

__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ partials, int nblk, int C, double count, float eps,
                                   float momentum, float* __restrict__ mean, float* __restrict__ invstd,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   long long* __restrict__ num_batches_tracked) {
  double s1, s2;
  bn_sum_partials(partials, nblk, C, s1, s2);
  const int c = threadIdx.x;
  if (c < C) {
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean != nullptr) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * m);
      running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
  }
  if (c == 0 && num_batches_tracked != nullptr) *num_batches_tracked += 1;
}

struct PoolGeom {
  int BT, T, H, W, C, p, Hp, Wp;
  int64_t osB, osT, osP, osC;  // element strides of the pooled output / its gradient
  int lc4;                     // log2(C / 4)
  int contig;                  // the pooled tensor is plain channels-last [BT][Hp][Wp][C]: element (position, c) sits at position * C + c
};

// Indexing of the three kernels below (round 4).  They were written as one flat grid-stride loop whose every element took its (bt, y, x, c)
// apart with 64-bit divisions by runtime sizes: ~1000 instructions per 16-byte element in bwd_dx, 3400 per trip of bwd_reduce -- at 51 M
// elements the vector work, not HBM, set their time (0.7 + 0.55 + 0.4 ms per step).  Now a workgroup walks ROWS (blockIdx-strided: bt and y
// come from scalar divisions once per row), a thread's channel block is the same for every element it ever touches (256 is a multiple of C / 4:
// the per-channel constants are loaded once), x is a shift, the pool size is a template constant, and offsets inside a row are 32-bit.

__device__ __forceinline__ float act_fwd(float v, int act) { return act == ACT_TANH ? tanhf(v) : (v > 0.f ? v : 0.01f * v); }
__device__ __forceinline__ float act_bwd_from_out(float out, int act) {
  return act == ACT_TANH ? 1.f - out * out : (out > 0.f ? 1.f : 0.01f);
}

// every thread handles 4 consecutive channels of one (pooled) position: float4 traffic on the channels-last side; a workgroup walks pooled rows
template <int P>
__global__ __launch_bounds__(256) void bn_pool_act_fwd_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ out, unsigned char* __restrict__ argmax,
                                                              unsigned short* __restrict__ out16, unsigned short* __restrict__ out_bf16,
                                                              PoolGeom g, int act) {
  const int C = g.C, C4 = C >> 2, tid = threadIdx.x;
  const int c = (tid & (C4 - 1)) * 4;                 // the same for every element of this thread (256 % C4 == 0)
  const float4 ga = *reinterpret_cast<const float4*>(gamma + c), is = *reinterpret_cast<const float4*>(invstd + c);
  const float4 be = *reinterpret_cast<const float4*>(beta + c), mu = *reinterpret_cast<const float4*>(mean + c);
  const float sc[4] = {ga.x * is.x, ga.y * is.y, ga.z * is.z, ga.w * is.w};
  const float sh[4] = {be.x - mu.x * sc[0], be.y - mu.y * sc[1], be.z - mu.z * sc[2], be.w - mu.w * sc[3]};
  const int rows = g.BT * g.Hp, n = g.Wp * C4;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {      // block-uniform: scalar arithmetic
    const int bt = row / g.Hp, py = row - bt * g.Hp, b = bt / g.T, t = bt - b * g.T;
    const float* yrow = y + ((int64_t)bt * g.H + (int64_t)py * P) * g.W * C + c;
    const int64_t prow = (int64_t)row * g.Wp * C + c;              // channels-last pooled index of (row, column 0, c)
    float* orow = out + (b * g.osB + t * g.osT + (int64_t)py * g.Wp * g.osP + c * g.osC);
    for (int e = tid; e < n; e += 256) {
      const int px = e >> g.lc4;
      const float* yp = yrow + px * P * C;
      float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      int bi[4] = {0, 0, 0, 0};
      float4 v4[P * P];
#pragma unroll
      for (int dy = 0; dy < P; ++dy)
#pragma unroll
        for (int dx = 0; dx < P; ++dx) v4[dy * P + dx] = *reinterpret_cast<const float4*>(yp + (dy * g.W + dx) * C);
#pragma unroll
      for (int q = 0; q < P * P; ++q) {
        const float v[4] = {v4[q].x * sc[0] + sh[0], v4[q].y * sc[1] + sh[1], v4[q].z * sc[2] + sh[2], v4[q].w * sc[3] + sh[3]};
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2)
          if (v[e2] > best[e2] || (v[e2] != v[e2] && best[e2] == best[e2])) { best[e2] = v[e2]; bi[e2] = q; }      // q = dy * P + dx
      }
      float* op = orow + px * g.osP;
      const float a4[4] = {act_fwd(best[0], act), act_fwd(best[1], act), act_fwd(best[2], act), act_fwd(best[3], act)};
      if (g.osC == 1) {
        *reinterpret_cast<float4*>(op) = make_float4(a4[0], a4[1], a4[2], a4[3]);
      } else {
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) op[e2 * g.osC] = a4[e2];
      }
      // the next Conv3d's forward MFMA rounds this activation to IEEE half when it stages it: done here once instead
      // (same rounding, same value), so that its halo becomes a plain copy of half the bytes
      const int64_t pi = prow + (int64_t)px * C;
      if (out16 != nullptr) *reinterpret_cast<uint2*>(out16 + pi) = make_uint2(pack2<2>(a4[0], a4[1]), pack2<2>(a4[2], a4[3]));
      if (out_bf16 != nullptr) *reinterpret_cast<uint2*>(out_bf16 + pi) = make_uint2(pack2<0>(a4[0], a4[1]), pack2<0>(a4[2], a4[3]));
      if (argmax != nullptr) *reinterpret_cast<uchar4*>(argmax + pi) = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
    }
  }
}

__device__ __forceinline__ void load4_strided(const float* p, int64_t stride, float v[4]) {
  if (stride == 1) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = p[e * stride];
  }
}

__global__ __launch_bounds__(256) void bn_pool_act_bwd_reduce_kernel(
    const float* __restrict__ dout, const float* __restrict__ out, const unsigned char* __restrict__ argmax,
    const float* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma_inv, float* __restrict__ partials, PoolGeom g, int act, int64_t rows_per_block) {
  // rows = pooled positions; thread -> (4 channels, row phase)
  // gamma_inv != null (LeakyReLU layers, beta known to the finalize kernel): the normalised input at the argmax is not
  // gathered from y -- a scattered read of nearly every line of the largest tensors of the step -- but recovered from
  // the pooled output, z = leaky^-1(out) = gamma * xhat + beta: the sum S2' = sum dz * z is accumulated here and turned
  // into sum dz * xhat = (S2' - beta * S1) / gamma by bn_bwd_finalize_kernel.  Channels with |gamma| < BN_INV_MIN_GAMMA
  // (xhat not recoverable) keep the gather.
  __shared__ float4 red[2][256];
  const int tid = threadIdx.x, C = g.C, C4 = C >> 2, c = (tid % C4) * 4, ph = tid / C4, nph = 256 / C4;
  const int64_t rows = (int64_t)g.BT * g.Hp * g.Wp;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  const float4 mu4 = *reinterpret_cast<const float4*>(mean + c), is4 = *reinterpret_cast<const float4*>(invstd + c);
  const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, is[4] = {is4.x, is4.y, is4.z, is4.w};
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  bool inv[4] = {false, false, false, false};
  if (gamma_inv != nullptr)
#pragma unroll
    for (int e = 0; e < 4; ++e) inv[e] = fabsf(gamma_inv[c + e]) >= BN_INV_MIN_GAMMA;
  const bool need_geo = !g.contig || !(inv[0] && inv[1] && inv[2] && inv[3]);      // strided pooled tensors, or channels that gather from y
  // four positions per trip, all their loads issued before the first use: with one position per trip a thread had one
  // dependent round trip to HBM in flight (the kernel streamed its 1.6 GB per step at 2.5 TB/s)
  constexpr int UB = 4;
  for (int64_t pos0 = r0 + ph; pos0 < r1; pos0 += (int64_t)UB * nph) {
    float dv[UB][4], ov[UB][4];
    uchar4 bi4[UB];
    int pxs[UB], pys[UB], bts[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int64_t pos = min(pos0 + (int64_t)u * nph, r1 - 1);      // clamped: the tail's duplicates are masked below
      int px = 0, py = 0, bt = 0;
      int64_t oi = pos * C + c;                                     // plain channels-last pooled tensors: no coordinates needed
      if (need_geo) {                                                // 32-bit divisions (the entry point checks rows < 2^31)
        const unsigned up = (unsigned)pos, q1 = up / (unsigned)g.Wp;
        px = (int)(up - q1 * (unsigned)g.Wp);
        bt = (int)(q1 / (unsigned)g.Hp);
        py = (int)(q1 - (unsigned)bt * (unsigned)g.Hp);
        const int b = bt / g.T, t = bt - b * g.T;
        oi = b * g.osB + t * g.osT + ((int64_t)py * g.Wp + px) * g.osP + c * g.osC;
      }
      load4_strided(dout + oi, g.osC, dv[u]);
      load4_strided(out + oi, g.osC, ov[u]);
      bi4[u] = make_uchar4(0, 0, 0, 0);
      if (argmax != nullptr) bi4[u] = *reinterpret_cast<const uchar4*>(argmax + pos * C + c);
      pxs[u] = px; pys[u] = py; bts[u] = bt;
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      if (pos0 + (int64_t)u * nph >= r1) continue;
      const int bi[4] = {bi4[u].x, bi4[u].y, bi4[u].z, bi4[u].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gg = dv[u][e] * act_bwd_from_out(ov[u][e], act);
        float xh;
        if (inv[e]) {
          xh = ov[u][e] > 0.f ? ov[u][e] : ov[u][e] * 100.f;          // z = LeakyReLU(0.01)^-1 (out)
        } else {
          const int iy = pys[u] * g.p + bi[e] / g.p, ix = pxs[u] * g.p + bi[e] % g.p;
          xh = (y[(((int64_t)bts[u] * g.H + iy) * g.W + ix) * C + c + e] - mu[e]) * is[e];
        }
        s1[e] += gg;
        s2[e] += gg * xh;
      }
    }
  }
  red[0][tid] = make_float4(s1[0], s1[1], s1[2], s1[3]);
  red[1][tid] = make_float4(s2[0], s2[1], s2[2], s2[3]);
  __syncthreads();
  if (tid < C4) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b2 = a;
    for (int p = 0; p < nph; ++p) {
      const float4 u = red[0][p * C4 + tid], w = red[1][p * C4 + tid];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
      b2.x += w.x; b2.y += w.y; b2.z += w.z; b2.w += w.w;
    }
    *reinterpret_cast<float4*>(partials + (int64_t)blockIdx.x * 2 * C + tid * 4) = a;
    *reinterpret_cast<float4*>(partials + (int64_t)blockIdx.x * 2 * C + C + tid * 4) = b2;
  }
}

// coef[0][c] = gamma*invstd, coef[1][c] = sum(g)/N, coef[2][c] = sum(g*xhat)/N
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ partials, int nblk, int C, double count,
                                       const float* __restrict__ gamma, const float* __restrict__ invstd,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                       float* __restrict__ coef, const float* __restrict__ beta_inv) {
  double s1, s2;
  bn_sum_partials(partials, nblk, C, s1, s2);
  const int c = threadIdx.x;
  if (c >= C) return;
  if (beta_inv != nullptr && fabsf(gamma[c]) >= BN_INV_MIN_GAMMA) s2 = (s2 - (double)beta_inv[c] * s1) / (double)gamma[c];
  if (dgamma != nullptr) {
    dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)s2;
    dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s1;
  }
  coef[c] = gamma[c] * invstd[c];
  coef[C + c] = (float)(s1 / count);
  coef[2 * C + c] = (float)(s2 / count);
}

template <bool DY16, int P>
__global__ __launch_bounds__(256) void bn_pool_act_bwd_dx_kernel(
    const float* __restrict__ dout, const float* __restrict__ out, const unsigned char* __restrict__ argmax,
    const float* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ coef, void* __restrict__ dy_, PoolGeom g, int act) {
  const int C = g.C, C4 = C >> 2, tid = threadIdx.x;
  const int c = (tid & (C4 - 1)) * 4;                 // the same for every element of this thread (256 % C4 == 0)
  const float4 mu = *reinterpret_cast<const float4*>(mean + c), is = *reinterpret_cast<const float4*>(invstd + c);
  const float4 k0 = *reinterpret_cast<const float4*>(coef + c), k1 = *reinterpret_cast<const float4*>(coef + C + c);
  const float4 k2 = *reinterpret_cast<const float4*>(coef + 2 * C + c);
  const int rows = g.BT * g.H, n = g.W * C4;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {      // block-uniform: scalar arithmetic
    const int bt = row / g.H, iy = row - bt * g.H, py = iy / P, b = bt / g.T, t = bt - b * g.T;
    const bool prow_ok = py < g.Hp;
    const int64_t rbase = (int64_t)row * g.W * C + c;              // element index of (row, column 0, c) in y / dy
    const int64_t pbase = ((int64_t)bt * g.Hp + py) * g.Wp * C + c;                                   // the same in the (channels-last) argmax
    const int64_t obase = b * g.osB + t * g.osT + (int64_t)py * g.Wp * g.osP + c * g.osC;             // ... in dout / out (their strides)
    const int herey = (iy - py * P) * P;
    for (int e = tid; e < n; e += 256) {
      const int ix = e >> g.lc4, px = ix / P;
      const float4 yv = *reinterpret_cast<const float4*>(y + rbase + (int64_t)ix * C);   // requested before the dependent argmax -> dout / out chain
      float gg[4] = {0.f, 0.f, 0.f, 0.f};
      if (prow_ok && px < g.Wp) {
        const int here = herey + (ix - px * P);
        uchar4 bi4 = make_uchar4(0, 0, 0, 0);
        if (argmax != nullptr) bi4 = *reinterpret_cast<const uchar4*>(argmax + pbase + (int64_t)px * C);
        const int bi[4] = {bi4.x, bi4.y, bi4.z, bi4.w};
        if (bi[0] == here || bi[1] == here || bi[2] == here || bi[3] == here) {
          const int64_t oi = obase + (int64_t)px * g.osP;
          float dv[4], ov[4];
          load4_strided(dout + oi, g.osC, dv);
          load4_strided(out + oi, g.osC, ov);
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2)
            if (bi[e2] == here) gg[e2] = dv[e2] * act_bwd_from_out(ov[e2], act);
        }
      }
      float4 o;
      o.x = k0.x * (gg[0] - k1.x - (yv.x - mu.x) * is.x * k2.x);
      o.y = k0.y * (gg[1] - k1.y - (yv.y - mu.y) * is.y * k2.y);
      o.z = k0.z * (gg[2] - k1.z - (yv.z - mu.z) * is.z * k2.z);
      o.w = k0.w * (gg[3] - k1.w - (yv.w - mu.w) * is.w * k2.w);
      // DY16: both consumers of dy (the input-gradient and the weight-gradient MFMA kernels) round it to bf16 when they stage
      // it -- rounded here once instead (bit-identical operands), written and re-read at half the bytes
      const int64_t di = rbase + (int64_t)ix * C;
      if constexpr (DY16) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(dy_) + di) = make_uint2(pack_bf2(o.x, o.y), pack_bf2(o.z, o.w));
      else *reinterpret_cast<float4*>(reinterpret_cast<float*>(dy_) + di) = o;
    }
  }
}

static int check_geom(const char* who, int B, int T, int H, int W, int C, int p) {
  MAAVSS_CHECK_ARG(B > 0 && T > 0 && H > 0 && W > 0, "%s: empty tensor", who);
  MAAVSS_CHECK_ARG(C >= 4 && C <= 64 && (C & (C - 1)) == 0, "%s: C must be a power of two in [4, 64] (got %d)", who, C);
  MAAVSS_CHECK_ARG(p >= 1 && p <= 3, "%s: pool must be 1, 2 or 3", who);
  MAAVSS_CHECK_ARG(H / p > 0 && W / p > 0, "%s: pooled size is zero", who);
  MAAVSS_CHECK_ARG((int64_t)B * T * H < (1LL << 31) && (int64_t)B * T * (H / p) * (W / p) < (1LL << 31), "%s: too many rows for the 32-bit row arithmetic", who);
  return MAAVSS_OK;
}

static PoolGeom make_geom(int B, int T, int H, int W, int C, int p, int64_t osB, int64_t osT, int64_t osP, int64_t osC) {
  PoolGeom g;
  g.BT = B * T; g.T = T; g.H = H; g.W = W; g.C = C; g.p = p; g.Hp = H / p; g.Wp = W / p;
  g.osB = osB; g.osT = osT; g.osP = osP; g.osC = osC;
  g.lc4 = 0;
  while ((4 << g.lc4) < C) ++g.lc4;
  g.contig = osC == 1 && osP == C && osT == (int64_t)g.Hp * g.Wp * C && osB == (int64_t)T * osT;
  return g;
}

extern "C" int maavss_bn_stats_nblk(int64_t rows) {
  int64_t n = (rows + 1023) / 1024;
  if (n > 512) n = 512;  // the finalize kernels walk the partial rows serially per channel
  if (n < 1) n = 1;
  return (int)n;
}

extern "C" int maavss_bn_stats(const float* y, float* partials, int64_t rows, int C, void* stream) {
  MAAVSS_CHECK_ARG(y && partials && rows > 0, "bn_stats: bad arguments");
  MAAVSS_CHECK_ARG(C >= 4 && C <= 64 && (C & (C - 1)) == 0, "bn_stats: C must be a power of two in [4, 64]");
  const int nblk = maavss_bn_stats_nblk(rows);
  const int rpb = cdiv(rows, nblk);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, y, partials, rows, C, rpb);
  MAAVSS_LAUNCH_CHECK("bn_stats_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_bn_finalize(const float* partials, int nblk, int C, double count, float eps, float momentum,
                                  float* mean, float* invstd, float* running_mean, float* running_var,
                                  void* num_batches_tracked, float* ws, void* stream) {
  MAAVSS_CHECK_ARG(partials && mean && invstd && nblk > 0 && C > 0 && C <= 64, "bn_finalize: bad arguments");
  if (nblk > 512 && ws != nullptr) {  // two-level reduction: 256 blocks fold the partial rows first (ws: 256*2*C floats)
    const int rpb = cdiv(nblk, 256), nb1 = cdiv(nblk, rpb);
    hipLaunchKernelGGL(bn_reduce_rows_kernel, dim3(nb1), dim3(256), 0, (hipStream_t)stream, partials, ws, nblk, 2 * C, rpb);
    MAAVSS_LAUNCH_CHECK("bn_reduce_rows_kernel");
    partials = ws;
    nblk = nb1;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, partials, nblk, C, count, eps, momentum,
                     mean, invstd, running_mean, running_var, (long long*)num_batches_tracked);
  MAAVSS_LAUNCH_CHECK("bn_finalize_kernel");
  return MAAVSS_OK;
}

// eval-mode BatchNorm: mean / invstd from the running statistics (model.eval(), torch.nn.BatchNorm semantics)
__global__ void bn_eval_stats_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var, float eps,
                                     float* __restrict__ mean, float* __restrict__ invstd, int C) {
  const int c = threadIdx.x;
  if (c < C) {
    mean[c] = running_mean[c];
    invstd[c] = rsqrtf(running_var[c] + eps);
  }
}

extern "C" int maavss_bn_eval_stats(const float* running_mean, const float* running_var, float eps, float* mean, float* invstd,
                                    int C, void* stream) {
  MAAVSS_CHECK_ARG(running_mean && running_var && mean && invstd && C > 0 && C <= 64, "bn_eval_stats: bad arguments");
  hipLaunchKernelGGL(bn_eval_stats_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, running_mean, running_var, eps, mean,
                     invstd, C);
  MAAVSS_LAUNCH_CHECK("bn_eval_stats_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_bn_pool_act_fwd(const float* y, const float* mean, const float* invstd, const float* gamma,
                                      const float* beta, float* out, void* argmax, int B, int T, int H, int W, int C,
                                      int pool, int act, int64_t os_b, int64_t os_t, int64_t os_p, int64_t os_c,
                                      void* out16, void* out_bf16, void* stream) {
  MAAVSS_CHECK_ARG(y && mean && invstd && gamma && beta && out, "bn_pool_act_fwd: null pointer");
  if (int rc = check_geom("bn_pool_act_fwd", B, T, H, W, C, pool)) return rc;
  MAAVSS_CHECK_ARG(pool == 1 || argmax != nullptr, "bn_pool_act_fwd: argmax buffer required when pool > 1");
  PoolGeom g = make_geom(B, T, H, W, C, pool, os_b, os_t, os_p, os_c);
  const dim3 grid(min(g.BT * g.Hp, 4096));      // a workgroup walks pooled rows
#define FWD_LAUNCH(P) hipLaunchKernelGGL(bn_pool_act_fwd_kernel<P>, grid, dim3(256), 0, (hipStream_t)stream, y, mean, invstd, gamma, beta, out, \
                                         (unsigned char*)argmax, (unsigned short*)out16, (unsigned short*)out_bf16, g, act)
  if (pool == 1) FWD_LAUNCH(1);
  else if (pool == 2) FWD_LAUNCH(2);
  else FWD_LAUNCH(3);
#undef FWD_LAUNCH
  MAAVSS_LAUNCH_CHECK("bn_pool_act_fwd_kernel");
  return MAAVSS_OK;
}

// ws: at least (2*C*nblk + 3*C) floats with nblk = maavss_bn_stats_nblk(B*T*Hp*Wp); coef = ws + 2*C*nblk
#define BN_DX_LAUNCH()                                                                                                        \
  {                                                                                                                           \
    const dim3 grid(min(g.BT * H, 4096));      /* a workgroup walks rows */                                                   \
    const unsigned char* am_ = (const unsigned char*)argmax;                                                                  \
    if (dy_bf16 && g.p == 1) hipLaunchKernelGGL((bn_pool_act_bwd_dx_kernel<true, 1>), grid, dim3(256), 0, st, dout, out, am_, y, mean, invstd, coef, dy, g, act);      \
    else if (dy_bf16 && g.p == 2) hipLaunchKernelGGL((bn_pool_act_bwd_dx_kernel<true, 2>), grid, dim3(256), 0, st, dout, out, am_, y, mean, invstd, coef, dy, g, act); \
    else if (dy_bf16) hipLaunchKernelGGL((bn_pool_act_bwd_dx_kernel<true, 3>), grid, dim3(256), 0, st, dout, out, am_, y, mean, invstd, coef, dy, g, act);             \
    else if (g.p == 1) hipLaunchKernelGGL((bn_pool_act_bwd_dx_kernel<false, 1>), grid, dim3(256), 0, st, dout, out, am_, y, mean, invstd, coef, dy, g, act);           \
    else if (g.p == 2) hipLaunchKernelGGL((bn_pool_act_bwd_dx_kernel<false, 2>), grid, dim3(256), 0, st, dout, out, am_, y, mean, invstd, coef, dy, g, act);           \
    else hipLaunchKernelGGL((bn_pool_act_bwd_dx_kernel<false, 3>), grid, dim3(256), 0, st, dout, out, am_, y, mean, invstd, coef, dy, g, act);                          \
    MAAVSS_LAUNCH_CHECK("bn_pool_act_bwd_dx_kernel");                                                                         \
  }

extern "C" int maavss_bn_pool_act_bwd(const float* dout, const float* out, const void* argmax, const float* y,
                                      const float* mean, const float* invstd, const float* gamma, const float* beta, void* dy,
                                      float* dgamma, float* dbeta, int accumulate, float* ws, int B, int T, int H, int W,
                                      int C, int pool, int act, int64_t os_b, int64_t os_t, int64_t os_p, int64_t os_c,
                                      int dy_bf16, void* stream) {
  MAAVSS_CHECK_ARG(dout && out && y && mean && invstd && gamma && ws, "bn_pool_act_bwd: null pointer");
  if (int rc = check_geom("bn_pool_act_bwd", B, T, H, W, C, pool)) return rc;
  MAAVSS_CHECK_ARG(pool == 1 || argmax != nullptr, "bn_pool_act_bwd: argmax buffer required when pool > 1");
  hipStream_t st = (hipStream_t)stream;
  PoolGeom g = make_geom(B, T, H, W, C, pool, os_b, os_t, os_p, os_c);
  const int64_t rows = (int64_t)g.BT * g.Hp * g.Wp;
  const int nblk = maavss_bn_stats_nblk(rows);
  float* coef = ws + (int64_t)2 * C * nblk;
  const bool inverse = beta != nullptr && act == ACT_LEAKY;   // tanh^-1 is ill-conditioned near +-1: those layers gather
  hipLaunchKernelGGL(bn_pool_act_bwd_reduce_kernel, dim3(nblk), dim3(256), 0, st, dout, out, (const unsigned char*)argmax,
                     y, mean, invstd, inverse ? gamma : nullptr, ws, g, act, (int64_t)cdiv(rows, nblk));
  MAAVSS_LAUNCH_CHECK("bn_pool_act_bwd_reduce_kernel");
  const double count = (double)g.BT * H * W;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(1024), 0, st, ws, nblk, C, count, gamma, invstd, dgamma, dbeta,
                     accumulate, coef, inverse ? beta : nullptr);
  MAAVSS_LAUNCH_CHECK("bn_bwd_finalize_kernel");
  if (dy == nullptr) return MAAVSS_OK;   // the consumer applies the coefficients itself (maavss_conv3d_c1_wgrad_bn)
  BN_DX_LAUNCH()
  return MAAVSS_OK;
}

// ---- split forms for cross-rank (global-batch) BatchNorm statistics: the per-channel sums leave the device-side
// reduction as doubles [2][C] (+ the element count in slot 2C), the host all-reduces them over the data-parallel group
// (RCCL), and the finish kernels take the reduced sums.  Single-device semantics of avse_model_final.py:35,40,...,103
// (one BatchNorm over the whole batch) when the batch is sharded over GPUs.
__global__ __launch_bounds__(1024) void bn_sums_kernel(const float* __restrict__ partials, int nblk, int C, double count,
                                                       double* __restrict__ sums) {
  double s1, s2;
  bn_sum_partials(partials, nblk, C, s1, s2);
  const int c = threadIdx.x;
  if (c < C) {
    sums[c] = s1;
    sums[C + c] = s2;
  }
  if (c == 0) sums[2 * C] = count;
}

__global__ void bn_finalize_sums_kernel(const double* __restrict__ sums, int C, float eps, float momentum, float* __restrict__ mean,
                                        float* __restrict__ invstd, float* __restrict__ running_mean,
                                        float* __restrict__ running_var, long long* __restrict__ num_batches_tracked) {
  const int c = threadIdx.x;
  const double count = sums[2 * C];
  if (c < C) {
    const double m = sums[c] / count;
    double var = sums[C + c] / count - m * m;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean != nullptr) {
      const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
      running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * m);
      running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
  }
  if (c == 0 && num_batches_tracked != nullptr) *num_batches_tracked += 1;
}

// dgamma / dbeta from THIS rank's sums (the gradient all-reduce adds the ranks), dx coefficients from the global ones
__global__ void bn_bwd_finalize_sums_kernel(const double* __restrict__ local, const double* __restrict__ global, int C,
                                            const float* __restrict__ gamma, const float* __restrict__ invstd,
                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                            float* __restrict__ coef, const float* __restrict__ beta_inv) {
  const int c = threadIdx.x;
  if (c >= C) return;
  const bool inv = beta_inv != nullptr && fabsf(gamma[c]) >= BN_INV_MIN_GAMMA;
  double l1 = local[c], l2 = local[C + c], g1 = global[c], g2 = global[C + c];
  if (inv) {
    l2 = (l2 - (double)beta_inv[c] * l1) / (double)gamma[c];
    g2 = (g2 - (double)beta_inv[c] * g1) / (double)gamma[c];
  }
  if (dgamma != nullptr) {
    dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)l2;
    dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)l1;
  }
  const double count = global[2 * C];
  coef[c] = gamma[c] * invstd[c];
  coef[C + c] = (float)(g1 / count);
  coef[2 * C + c] = (float)(g2 / count);
}

extern "C" int maavss_bn_partials_to_sums(const float* partials, int nblk, int C, double count, double* sums, float* ws,
                                          void* stream) {
  MAAVSS_CHECK_ARG(partials && sums && nblk > 0 && C > 0 && C <= 64 && count > 0, "bn_partials_to_sums: bad arguments");
  if (nblk > 512 && ws != nullptr) {
    const int rpb = cdiv(nblk, 256), nb1 = cdiv(nblk, rpb);
    hipLaunchKernelGGL(bn_reduce_rows_kernel, dim3(nb1), dim3(256), 0, (hipStream_t)stream, partials, ws, nblk, 2 * C, rpb);
    MAAVSS_LAUNCH_CHECK("bn_reduce_rows_kernel");
    partials = ws;
    nblk = nb1;
  }
  hipLaunchKernelGGL(bn_sums_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, partials, nblk, C, count, sums);
  MAAVSS_LAUNCH_CHECK("bn_sums_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_bn_finalize_sums(const double* sums, int C, float eps, float momentum, float* mean, float* invstd,
                                       float* running_mean, float* running_var, void* num_batches_tracked, void* stream) {
  MAAVSS_CHECK_ARG(sums && mean && invstd && C > 0 && C <= 64, "bn_finalize_sums: bad arguments");
  hipLaunchKernelGGL(bn_finalize_sums_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, C, eps, momentum, mean, invstd,
                     running_mean, running_var, (long long*)num_batches_tracked);
  MAAVSS_LAUNCH_CHECK("bn_finalize_sums_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_bn_pool_act_bwd_sums(const float* dout, const float* out, const void* argmax, const float* y,
                                           const float* mean, const float* invstd, const float* gamma, const float* beta,
                                           float* ws, double* sums, int B, int T, int H, int W, int C, int pool, int act,
                                           int64_t os_b, int64_t os_t, int64_t os_p, int64_t os_c, void* stream) {
  MAAVSS_CHECK_ARG(dout && out && y && mean && invstd && gamma && ws && sums, "bn_pool_act_bwd_sums: null pointer");
  if (int rc = check_geom("bn_pool_act_bwd_sums", B, T, H, W, C, pool)) return rc;
  MAAVSS_CHECK_ARG(pool == 1 || argmax != nullptr, "bn_pool_act_bwd_sums: argmax buffer required when pool > 1");
  hipStream_t st = (hipStream_t)stream;
  PoolGeom g = make_geom(B, T, H, W, C, pool, os_b, os_t, os_p, os_c);
  const int64_t rows = (int64_t)g.BT * g.Hp * g.Wp;
  const int nblk = maavss_bn_stats_nblk(rows);
  const bool inverse = beta != nullptr && act == ACT_LEAKY;
  hipLaunchKernelGGL(bn_pool_act_bwd_reduce_kernel, dim3(nblk), dim3(256), 0, st, dout, out, (const unsigned char*)argmax,
                     y, mean, invstd, inverse ? gamma : nullptr, ws, g, act, (int64_t)cdiv(rows, nblk));
  MAAVSS_LAUNCH_CHECK("bn_pool_act_bwd_reduce_kernel");
  hipLaunchKernelGGL(bn_sums_kernel, dim3(1), dim3(1024), 0, st, ws, nblk, C, (double)g.BT * H * W, sums);
  MAAVSS_LAUNCH_CHECK("bn_sums_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_bn_pool_act_bwd_finish(const float* dout, const float* out, const void* argmax, const float* y,
                                             const float* mean, const float* invstd, const float* gamma, const float* beta,
                                             void* dy, float* dgamma, float* dbeta, int accumulate, const double* sums_local,
                                             const double* sums_global, float* coef, int B, int T, int H, int W, int C, int pool,
                                             int act, int64_t os_b, int64_t os_t, int64_t os_p, int64_t os_c, int dy_bf16, void* stream) {
  MAAVSS_CHECK_ARG(dout && out && y && mean && invstd && gamma && sums_local && sums_global && coef,
                   "bn_pool_act_bwd_finish: null pointer");
  if (int rc = check_geom("bn_pool_act_bwd_finish", B, T, H, W, C, pool)) return rc;
  hipStream_t st = (hipStream_t)stream;
  PoolGeom g = make_geom(B, T, H, W, C, pool, os_b, os_t, os_p, os_c);
  const bool inverse = beta != nullptr && act == ACT_LEAKY;
  hipLaunchKernelGGL(bn_bwd_finalize_sums_kernel, dim3(1), dim3(64), 0, st, sums_local, sums_global, C, gamma, invstd, dgamma,
                     dbeta, accumulate, coef, inverse ? beta : nullptr);
  MAAVSS_LAUNCH_CHECK("bn_bwd_finalize_sums_kernel");
  if (dy == nullptr) return MAAVSS_OK;
  BN_DX_LAUNCH()
  return MAAVSS_OK;
}
