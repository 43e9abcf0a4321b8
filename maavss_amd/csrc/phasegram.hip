// K20: utilities.video_phasegram (utilities.py:206-228; called on the attention frames by
// train_av_net.py:122-125) -- fft2 -> fftshift (over all four axes, as the reference calls it) -> angle -> flatten -> cumulative sum / (2 pi N) [or (angle + pi) / 2 pi]
// -> temporal difference with a zero first row -> division by the maximum magnitude of the WHOLE batch tensor.
//   pass 1, one workgroup per frame: the P x P frame is scattered bit-reversed (rows and columns) into LDS as complex
//           numbers, in-place radix-2 DIT FFTs run along the rows and then along the columns (twiddles from an LDS table),
//           the four self-conjugate bins get an exactly zero imaginary part (as a real-input transform produces: their
//           angle is 0 or +pi, never -pi by rounding noise), the shifted phase is scanned over the P*P bins in one block
//           scan (16 consecutive bins per thread) and written as p[frame][bin];
//   pass 2: out = p[t] - p[t-1] (zero row at t = 0) and the global max |out| by atomicMax on the float bits;
//   pass 3: out *= 1 / max.
// P in {32, 64} (the reference's --p_size default is 64); 4 * P*P bytes in and out per frame: HBM / latency-bound.
#include "common.h"

template <int P>
__global__ __launch_bounds__(256) void phasegram_frame_kernel(const float* __restrict__ frames, float* __restrict__ p_out,
                                                              int B, int T, int cumulative) {
  constexpr int LOG2P = P == 32 ? 5 : 6;
  constexpr int N = P * P;
  constexpr int PER = N / 256;                 // consecutive bins per thread in the scan (4 or 16)
  __shared__ float2 x[P][P + 1];               // +1: column passes walk a column without bank conflicts
  __shared__ float2 tw[P / 2];
  __shared__ float wsum[4];
  const int tid = threadIdx.x;
  // torch.fft.fftshift without `dim` (utilities.py:210) rolls EVERY dimension by n // 2 -- the batch and the frame axis
  // too: output (b, t) is the spectrum of input frame ((b - B/2) mod B, (t - T/2) mod T).  Reference behaviour, kept.
  const int ob = blockIdx.x / T, ot = blockIdx.x % T;
  const int ib = (ob - B / 2 + B) % B, it = (ot - T / 2 + T) % T;
  const float* f = frames + ((int64_t)ib * T + it) * N;
  if (tid < P / 2) {
    float s, c;
    sincospif(-2.0f * (float)tid / (float)P, &s, &c);
    tw[tid] = make_float2(c, s);
  }
  for (int i = tid; i < N; i += 256) {
    const int r = i / P, c = i % P;
    const int rr = (int)(__brev((unsigned)r) >> (32 - LOG2P)), cr = (int)(__brev((unsigned)c) >> (32 - LOG2P));
    x[rr][cr] = make_float2(f[i], 0.f);
  }
  __syncthreads();
  // rows: butterfly j of row r in stage s pairs columns (base + k, base + k + half)
  for (int s = 0; s < LOG2P; ++s) {
    const int half = 1 << s;
    for (int i = tid; i < N / 2; i += 256) {
      const int r = i / (P / 2), j = i % (P / 2);
      const int k = j & (half - 1), base = (j >> s) << (s + 1);
      const float2 w = tw[k * (P / (2 * half))];
      const float2 a = x[r][base + k], b = x[r][base + k + half];
      const float2 t = make_float2(b.x * w.x - b.y * w.y, b.x * w.y + b.y * w.x);
      x[r][base + k] = make_float2(a.x + t.x, a.y + t.y);
      x[r][base + k + half] = make_float2(a.x - t.x, a.y - t.y);
    }
    __syncthreads();
  }
  // columns
  for (int s = 0; s < LOG2P; ++s) {
    const int half = 1 << s;
    for (int i = tid; i < N / 2; i += 256) {
      const int c = i % P, j = i / P;
      const int k = j & (half - 1), base = (j >> s) << (s + 1);
      const float2 w = tw[k * (P / (2 * half))];
      const float2 a = x[base + k][c], b = x[base + k + half][c];
      const float2 t = make_float2(b.x * w.x - b.y * w.y, b.x * w.y + b.y * w.x);
      x[base + k][c] = make_float2(a.x + t.x, a.y + t.y);
      x[base + k + half][c] = make_float2(a.x - t.x, a.y - t.y);
    }
    __syncthreads();
  }
  if (tid < 4) x[(tid >> 1) * (P / 2)][(tid & 1) * (P / 2)].y = 0.f;     // self-conjugate bins are real
  __syncthreads();
  // shifted phase, flattened row-major: bin (u, v) of the output = spectrum[(u + P/2) % P][(v + P/2) % P]
  float v[PER];
  float run = 0.f;
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int i = tid * PER + e, u = i / P, vv = i % P;
    const float2 z = x[(u + P / 2) % P][(vv + P / 2) % P];
    const float ph = atan2f(z.y, z.x);
    if (cumulative) {
      run += ph;
      v[e] = run;
    } else {
      v[e] = (ph + 3.14159265358979323846f) * (1.0f / (2.0f * 3.14159265358979323846f));
    }
  }
  float* out = p_out + (int64_t)blockIdx.x * N + tid * PER;
  if (cumulative) {
    // exclusive scan of the per-thread totals: inclusive wave scan by lane shifts, then the wave totals
    const int lane = tid & 63, wv = tid >> 6;
    float inc = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float n = __shfl_up(inc, o, 64);
      if (lane >= o) inc += n;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    float offset = inc - run;
    for (int w2 = 0; w2 < wv; ++w2) offset += wsum[w2];
    const float scale = 1.0f / (2.0f * 3.14159265358979323846f * (float)N);
#pragma unroll
    for (int e = 0; e < PER; ++e) out[e] = (v[e] + offset) * scale;
  } else {
#pragma unroll
    for (int e = 0; e < PER; ++e) out[e] = v[e];
  }
}

// out[b][t][i] = p[b][t][i] - p[b][t-1][i] (0 at t = 0) when diff, else p; absmax over everything
__global__ __launch_bounds__(256) void phasegram_diff_kernel(const float* __restrict__ p, float* __restrict__ out, int T, int N,
                                                             int64_t total, int diff, float* __restrict__ absmax) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    float v = p[i];
    if (diff) {
      const int t = (int)((i / N) % T);
      v = t == 0 ? 0.f : v - p[i - N];
    }
    out[i] = v;
    m = fmaxf(m, fabsf(v));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax((unsigned int*)absmax, __float_as_uint(m));   // m >= 0: bit order = value order
}

__global__ __launch_bounds__(256) void phasegram_scale_kernel(float* __restrict__ out, int64_t total, const float* __restrict__ absmax) {
  const float s = 1.0f / *absmax;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) out[i] *= s;
}

extern "C" int maavss_video_phasegram(const float* frames, int64_t batch, int T, int P, int diff, int cumulative, int normalize,
                                      float* p_ws, float* absmax_ws, float* out, void* stream) {
  MAAVSS_CHECK_ARG(frames && p_ws && absmax_ws && out && batch > 0 && T > 0, "video_phasegram: bad arguments");
  MAAVSS_CHECK_ARG(P == 32 || P == 64, "video_phasegram: frame size must be 32 or 64 (got %d)", P);
  hipStream_t st = (hipStream_t)stream;
  const int nframes = (int)(batch * T);
  if (P == 32) hipLaunchKernelGGL(phasegram_frame_kernel<32>, dim3(nframes), dim3(256), 0, st, frames, p_ws, (int)batch, T, cumulative);
  else hipLaunchKernelGGL(phasegram_frame_kernel<64>, dim3(nframes), dim3(256), 0, st, frames, p_ws, (int)batch, T, cumulative);
  MAAVSS_LAUNCH_CHECK("phasegram_frame_kernel");
  const int64_t total = (int64_t)nframes * P * P;
  int grid = (int)((total + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipMemsetAsync(absmax_ws, 0, sizeof(float), st);
  hipLaunchKernelGGL(phasegram_diff_kernel, dim3(grid), dim3(256), 0, st, p_ws, out, T, P * P, total, diff, absmax_ws);
  MAAVSS_LAUNCH_CHECK("phasegram_diff_kernel");
  if (normalize) {
    hipLaunchKernelGGL(phasegram_scale_kernel, dim3(grid), dim3(256), 0, st, out, total, absmax_ws);
    MAAVSS_LAUNCH_CHECK("phasegram_scale_kernel");
  }
  return MAAVSS_OK;
}

// ---- bilinear resize of the attention frames in front of the phasegram (utilities.py:208-209:
// torchvision.transforms.functional.resize on a tensor = torch.nn.functional.interpolate(mode="bilinear",
// align_corners=False), no antialiasing in the torchvision of the reference's time).  One thread per output pixel.
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t n,
                                                              int H, int W, int h, int w) {
  const int64_t total = n * h * w;
  const float sy = (float)H / (float)h, sx = (float)W / (float)w;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ox = (int)(i % w), oy = (int)((i / w) % h);
    const int64_t img = i / ((int64_t)w * h);
    float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float* p = in + img * H * W;
    const float top = p[(int64_t)y0 * W + x0] * (1.f - lx) + p[(int64_t)y0 * W + x1] * lx;
    const float bot = p[(int64_t)y1 * W + x0] * (1.f - lx) + p[(int64_t)y1 * W + x1] * lx;
    out[i] = top * (1.f - ly) + bot * ly;
  }
}

extern "C" int maavss_resize_bilinear(const float* in, float* out, int64_t n, int H, int W, int h, int w, void* stream) {
  MAAVSS_CHECK_ARG(in && out && n > 0 && H > 0 && W > 0 && h > 0 && w > 0, "resize_bilinear: bad arguments");
  const int64_t total = n * h * w;
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, out, n, H, W, h, w);
  MAAVSS_LAUNCH_CHECK("resize_bilinear_kernel");
  return MAAVSS_OK;
}
