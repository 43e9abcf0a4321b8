// K12: the bidirectional LSTM of the fusion network (reference avse_model_final.py:132-133, 239-242):
// hidden 256, no bias, batch_first, zero initial state; the "time" axis is the L = latent_channels (16)
// steps.  The input projection X.W_ih^T for all steps and both directions is one GEMM (gemm.hip); the
// recurrence runs one small launch per step (both directions in one launch, blockIdx.y), each block owning
// JT hidden units x all 4 gates with its W_hh slice and the previous hidden state in LDS (staged with all of a
// thread's float4 loads in flight at once).  Latency-bound, < 1 % of the step's FLOPs.  Gate order i, f, g, o (torch.nn.LSTM).  f32 VALU arithmetic in both modes.
//
// Buffers (f32):
//   gx   [B][L][2][4][H]   input projections (forward) / pre-activation gate gradients (backward, in place
//                          of a separate buffer the caller passes `dgx` of the same shape)
//   av   [B][L][2*H]       outputs  (h_fwd(t) | h_bwd(t))
//   hp   [B][L][2][H]      hidden state that ENTERED step t (zero at each direction's first step)
//   gs   [B][L][2][4][H]   post-activation gates i,f,g,o
//   cs   [B][L][2][H]      cell state after step t
#include "common.h"

#define LH 256
#define JT 8
#define BB 32

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + __expf(-v)); }

__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(const float* __restrict__ gx, const float* __restrict__ whh_f,
                                                            const float* __restrict__ whh_b, float* __restrict__ av,
                                                            float* __restrict__ hp, float* __restrict__ gs,
                                                            float* __restrict__ cs, int B, int L, int s) {
  constexpr int KC = 128;                 // K chunk staged in LDS
  __shared__ float wl[KC][4 * JT];        // W_hh^T slice: [k][gate*JT + jj]
  __shared__ float hl[BB][KC + 1];
  const int d = blockIdx.y, j0 = blockIdx.x * JT, tid = threadIdx.x;
  const int t = d == 0 ? s : L - 1 - s, tprev = d == 0 ? t - 1 : t + 1;
  const float* whh = d == 0 ? whh_f : whh_b;
  const bool first = s == 0;
  const int bb = tid / JT, jj = tid % JT, j = j0 + jj;
  for (int b0 = 0; b0 < B; b0 += BB) {
    const int b = b0 + bb;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int kc = 0; kc < LH; kc += KC) {
      // staging: this thread's 4 + 4 float4 are all requested before the first is used (the scalar one-load-in-flight
      // loops made a step 64 dependent memory round trips: 22 us for 17 MFLOP)
      float4 wv[4], hv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = tid + u * 256, row = i / (KC / 4), k4 = (i % (KC / 4)) * 4, gate = row / JT, q = row % JT;
        wv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!first) wv[u] = *reinterpret_cast<const float4*>(whh + (int64_t)(gate * LH + j0 + q) * LH + kc + k4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = tid + u * 256, r = i / (KC / 4), k4 = (i % (KC / 4)) * 4, br = b0 + r;
        hv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (br < B && !first) hv[u] = *reinterpret_cast<const float4*>(av + ((int64_t)br * L + tprev) * 2 * LH + d * LH + kc + k4);
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = tid + u * 256, row = i / (KC / 4), k4 = (i % (KC / 4)) * 4;
        wl[k4][row] = wv[u].x; wl[k4 + 1][row] = wv[u].y; wl[k4 + 2][row] = wv[u].z; wl[k4 + 3][row] = wv[u].w;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = tid + u * 256, r = i / (KC / 4), k4 = (i % (KC / 4)) * 4, br = b0 + r;
        hl[r][k4] = hv[u].x; hl[r][k4 + 1] = hv[u].y; hl[r][k4 + 2] = hv[u].z; hl[r][k4 + 3] = hv[u].w;
        if (br < B && j0 == 0) *reinterpret_cast<float4*>(hp + (((int64_t)br * L + t) * 2 + d) * LH + kc + k4) = hv[u];
      }
      __syncthreads();
      if (!first) {
#pragma unroll 8
        for (int k = 0; k < KC; ++k) {
          const float h = hl[bb][k];
          a0 = fmaf(h, wl[k][jj], a0);
          a1 = fmaf(h, wl[k][JT + jj], a1);
          a2 = fmaf(h, wl[k][2 * JT + jj], a2);
          a3 = fmaf(h, wl[k][3 * JT + jj], a3);
        }
      }
    }
    if (b < B) {
      const float* gxp = gx + (((int64_t)b * L + t) * 2 + d) * 4 * LH;
      a0 += gxp[j]; a1 += gxp[LH + j]; a2 += gxp[2 * LH + j]; a3 += gxp[3 * LH + j];
      const float ig = sigmoidf_(a0), fg = sigmoidf_(a1), gg = tanhf(a2), og = sigmoidf_(a3);
      const int64_t ci = (((int64_t)b * L + t) * 2 + d) * LH + j;
      const float cprev = first ? 0.f : cs[(((int64_t)b * L + tprev) * 2 + d) * LH + j];
      const float c = fg * cprev + ig * gg;
      const float h = og * tanhf(c);
      cs[ci] = c;
      float* gsp = gs + (((int64_t)b * L + t) * 2 + d) * 4 * LH;
      gsp[j] = ig; gsp[LH + j] = fg; gsp[2 * LH + j] = gg; gsp[3 * LH + j] = og;
      av[((int64_t)b * L + t) * 2 * LH + d * LH + j] = h;
    }
  }
}

// Backward step s (processing order L-1 .. 0).  dh(t) = dav(t) + dgates(t_next) . W_hh ;  then the pointwise
// LSTM backward for this block's hidden units, writing pre-activation gate gradients into dgx(t).
__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(const float* __restrict__ dav, const float* __restrict__ whh_f,
                                                            const float* __restrict__ whh_b, const float* __restrict__ gs,
                                                            const float* __restrict__ cs, float* __restrict__ dgx,
                                                            float* __restrict__ dc, int B, int L, int s) {
  __shared__ float wl[LH][JT + 1];   // W_hh[n-chunk][j0..j0+JT)
  __shared__ float dl[BB][LH + 1];   // dgates(t_next)[b][n-chunk]
  const int d = blockIdx.y, j0 = blockIdx.x * JT, tid = threadIdx.x;
  const int t = d == 0 ? s : L - 1 - s;
  const int tnext = d == 0 ? t + 1 : t - 1, tprev = d == 0 ? t - 1 : t + 1;
  const bool last = s == L - 1, first = s == 0;
  const float* whh = d == 0 ? whh_f : whh_b;
  const int bb = tid / JT, jj = tid % JT, j = j0 + jj;
  for (int b0 = 0; b0 < B; b0 += BB) {
    const int b = b0 + bb;
    float dh = 0.f;
    if (!last) {
      for (int n0 = 0; n0 < 4 * LH; n0 += LH) {
        // staging: 2 + 8 float4 per thread, all requested before the first is used (was 160 dependent round trips per step)
        float4 wv[2], dv[8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int i = tid + u * 256, n = i / (JT / 4), q4 = (i % (JT / 4)) * 4;
          wv[u] = *reinterpret_cast<const float4*>(whh + (int64_t)(n0 + n) * LH + j0 + q4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = tid + u * 256, r = i / (LH / 4), n4 = (i % (LH / 4)) * 4;
          dv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (b0 + r < B) dv[u] = *reinterpret_cast<const float4*>(dgx + (((int64_t)(b0 + r) * L + tnext) * 2 + d) * 4 * LH + n0 + n4);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int i = tid + u * 256, n = i / (JT / 4), q4 = (i % (JT / 4)) * 4;
          wl[n][q4] = wv[u].x; wl[n][q4 + 1] = wv[u].y; wl[n][q4 + 2] = wv[u].z; wl[n][q4 + 3] = wv[u].w;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = tid + u * 256, r = i / (LH / 4), n4 = (i % (LH / 4)) * 4;
          dl[r][n4] = dv[u].x; dl[r][n4 + 1] = dv[u].y; dl[r][n4 + 2] = dv[u].z; dl[r][n4 + 3] = dv[u].w;
        }
        __syncthreads();
#pragma unroll 8
        for (int n = 0; n < LH; ++n) dh = fmaf(dl[bb][n], wl[n][jj], dh);
      }
    }
    if (b < B) {
      dh += dav[((int64_t)b * L + t) * 2 * LH + d * LH + j];
      const int64_t base = ((int64_t)b * L + t) * 2 + d;
      const float* gsp = gs + base * 4 * LH;
      const float ig = gsp[j], fg = gsp[LH + j], gg = gsp[2 * LH + j], og = gsp[3 * LH + j];
      const float c = cs[base * LH + j];
      const float cprev = first ? 0.f : cs[(((int64_t)b * L + tprev) * 2 + d) * LH + j];
      const float tc = tanhf(c);
      const int64_t dci = ((int64_t)d * B + b) * LH + j;
      const float dct = (last ? 0.f : dc[dci]) + dh * og * (1.f - tc * tc);
      float* o = dgx + base * 4 * LH;
      o[j] = dct * gg * ig * (1.f - ig);
      o[LH + j] = dct * cprev * fg * (1.f - fg);
      o[2 * LH + j] = dct * ig * (1.f - gg * gg);
      o[3 * LH + j] = dh * tc * og * (1.f - og);
      dc[dci] = dct * fg;
    }
  }
}

extern "C" int maavss_lstm_fwd(const float* gx, const float* whh_f, const float* whh_b, float* av, float* hp, float* gs,
                               float* cs, int B, int L, void* stream) {
  MAAVSS_CHECK_ARG(gx && whh_f && whh_b && av && hp && gs && cs, "lstm_fwd: null pointer");
  MAAVSS_CHECK_ARG(B > 0 && L > 0, "lstm_fwd: empty problem");
  for (int s = 0; s < L; ++s) {
    hipLaunchKernelGGL(lstm_fwd_step_kernel, dim3(LH / JT, 2), dim3(256), 0, (hipStream_t)stream, gx, whh_f, whh_b, av, hp,
                       gs, cs, B, L, s);
  }
  MAAVSS_LAUNCH_CHECK("lstm_fwd_step_kernel");
  return MAAVSS_OK;
}

// dc: scratch [2][B][H]
extern "C" int maavss_lstm_bwd(const float* dav, const float* whh_f, const float* whh_b, const float* gs, const float* cs,
                               float* dgx, float* dc, int B, int L, void* stream) {
  MAAVSS_CHECK_ARG(dav && whh_f && whh_b && gs && cs && dgx && dc, "lstm_bwd: null pointer");
  MAAVSS_CHECK_ARG(B > 0 && L > 0, "lstm_bwd: empty problem");
  for (int s = L - 1; s >= 0; --s) {
    hipLaunchKernelGGL(lstm_bwd_step_kernel, dim3(LH / JT, 2), dim3(256), 0, (hipStream_t)stream, dav, whh_f, whh_b, gs, cs,
                       dgx, dc, B, L, s);
  }
  MAAVSS_LAUNCH_CHECK("lstm_bwd_step_kernel");
  return MAAVSS_OK;
}
