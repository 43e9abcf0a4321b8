// K17: audio STFT (+ noise) -- replaces AV_Dataset.stft / gen_stft_example / add_noise
// (reference av_dataset.py:157-174, 217-220, 335-342).
//
// One wavefront per STFT frame: reflect-padded framing + (pre-scaled) periodic Hamming window are
// applied while the frame is loaded into LDS, a radix-2 Stockham FFT runs in LDS (ping-pong
// buffers, twiddles from an LDS table), and the one-sided bins are written straight into the
// [B, 2, T_a, F] (re/im plane, frame, bin) layout the model consumes, together with the noisy
// copy x = y + sigma * N(0,1).  HBM-bound: 4*L bytes in, 2 * 2*T_a*F*4 bytes out per clip.
#include "common.h"


// Round 3: (1) the waves of a workgroup are independent, so the FFT stages are ordered by the wave's own LDS queue (LDS operations of
// one wave complete in issue order) and a compiler-level wave barrier instead of ten workgroup barriers per frame; (2) a workgroup
// builds its twiddle table once and walks a grid-stride list of frame PAIRS; (3) two real frames share one complex FFT (z = a + i b,
// A[k] = (Z[k] + conj Z[N-k]) / 2, B[k] = (Z[k] - conj Z[N-k]) / 2i): half the butterflies and LDS passes per frame; (4) one Philox
// block serves two bins (its four normals: re / im of bins f and f + 64) instead of one.  profiles/r3_stft_bench.json.
// Round 4: the Stockham FFT runs in radix-8 / radix-4 passes (512 = 8.8.8, 256 = 4.4.4.4, 1024 = 8.8.4.4) with the butterflies in registers:
// three LDS round trips per frame pair instead of nine, 69 LDS instructions per lane instead of 180, ~270 vector instructions instead of ~900
// (the kernel is bound by its vector work, not by HBM: DESIGN.md); Box-Muller takes its angle through v_sin_f32 / v_cos_f32, whose argument
// is in revolutions -- exactly the uniform deviate -- instead of sincospif's software range reduction.  profiles/r4_stft_bench.json.
#define STFT_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                              __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 w) { return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x); }
// forward DFTs (kernel exp(-2 pi i r q / R)) of R points held in registers, in place
__device__ __forceinline__ void dft4(float2& a0, float2& a1, float2& a2, float2& a3) {
  const float2 s0 = cadd(a0, a2), d0 = csub(a0, a2), s1 = cadd(a1, a3), d1 = csub(a1, a3);
  a0 = cadd(s0, s1);
  a2 = csub(s0, s1);
  a1 = make_float2(d0.x + d1.y, d0.y - d1.x);      // d0 - i d1
  a3 = make_float2(d0.x - d1.y, d0.y + d1.x);      // d0 + i d1
}
template <int R>
__device__ __forceinline__ void dft_r(float2 (&a)[R]) {
  if constexpr (R == 4) {
    dft4(a[0], a[1], a[2], a[3]);
  } else {
    static_assert(R == 8, "radix 4 or 8");
    float2 e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6], o0 = a[1], o1 = a[3], o2 = a[5], o3 = a[7];
    dft4(e0, e1, e2, e3);
    dft4(o0, o1, o2, o3);
    constexpr float kS = 0.70710678118654752f;
    const float2 t1 = make_float2((o1.x + o1.y) * kS, (o1.y - o1.x) * kS);      // o1 (1 - i) / sqrt 2
    const float2 t2 = make_float2(o2.y, -o2.x);                                 // -i o2
    const float2 t3 = make_float2((o3.y - o3.x) * kS, -(o3.x + o3.y) * kS);     // o3 (-1 - i) / sqrt 2
    a[0] = cadd(e0, o0); a[4] = csub(e0, o0);
    a[1] = cadd(e1, t1); a[5] = csub(e1, t1);
    a[2] = cadd(e2, t2); a[6] = csub(e2, t2);
    a[3] = cadd(e3, t3); a[7] = csub(e3, t3);
  }
}
// One radix-R Stockham pass over N points (P = product of the radices of the earlier passes): butterfly i takes in[i + r N / R], multiplies by
// exp(-2 pi i r k / (R P)), k = i mod P, transforms, and writes out[(i - k) R + k + q P].  tw = the FULL table exp(-2 pi i q / N), q < N.
// When a pass is ONE butterfly per lane (N / R = 64) `out` may be `in`: the wave's reads are all issued before its first write and the LDS
// serves one wave's operations in issue order.
template <int N, int R, int P>
__device__ __forceinline__ void fft_pass(const float2* in, float2* out, const float2* __restrict__ tw, int lane) {
  constexpr int NB = N / R;
#pragma unroll
  for (int i0 = 0; i0 < NB; i0 += 64) {
    const int i = i0 + lane;
    if (NB < 64 && i >= NB) break;
    const int k = i & (P - 1);
    float2 a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = in[i + r * NB];
    if constexpr (P > 1) {
#pragma unroll
      for (int r = 1; r < R; ++r) a[r] = cmul(a[r], tw[r * k * (N / (R * P))]);
    }
    dft_r<R>(a);
    const int j = (i - k) * R + k;
#pragma unroll
    for (int q = 0; q < R; ++q) out[j + q * P] = a[q];
  }
}
// the whole transform.  512 and 256 points: every pass is one butterfly per lane -> IN PLACE in b0 (b1 unused: half the LDS, twice the
// workgroups per CU); 1024 points ping-pongs.  Returns the index (0 / 1) of the buffer that holds the result.
template <int N>
__device__ __forceinline__ int fft_forward(float2* b0, float2* b1, const float2* tw, int lane) {
  if constexpr (N == 512) {
    fft_pass<512, 8, 1>(b0, b0, tw, lane); STFT_WAVE_SYNC();
    fft_pass<512, 8, 8>(b0, b0, tw, lane); STFT_WAVE_SYNC();
    fft_pass<512, 8, 64>(b0, b0, tw, lane); STFT_WAVE_SYNC();
    return 0;
  } else if constexpr (N == 256) {
    fft_pass<256, 4, 1>(b0, b0, tw, lane); STFT_WAVE_SYNC();
    fft_pass<256, 4, 4>(b0, b0, tw, lane); STFT_WAVE_SYNC();
    fft_pass<256, 4, 16>(b0, b0, tw, lane); STFT_WAVE_SYNC();
    fft_pass<256, 4, 64>(b0, b0, tw, lane); STFT_WAVE_SYNC();
    return 0;
  } else {
    static_assert(N == 1024, "n_fft 256, 512 or 1024");
    fft_pass<1024, 8, 1>(b0, b1, tw, lane); STFT_WAVE_SYNC();
    fft_pass<1024, 8, 8>(b1, b0, tw, lane); STFT_WAVE_SYNC();
    fft_pass<1024, 4, 64>(b0, b1, tw, lane); STFT_WAVE_SYNC();
    fft_pass<1024, 4, 256>(b1, b0, tw, lane); STFT_WAVE_SYNC();
    return 0;
  }
}
// In-kernel noise, counter layout (Philox4x32-10 block = four normals): block ((fid * 8 + jp) * 64 + lane) = re / im of bins f0 = lane + 128 jp and
// f0 + 64 of frame fid, for the bins below n_fft / 2; the LAST bin (n_fft / 2, present when n_bins_out = n_fft / 2 + 1) of both frames of a pair
// takes block ((fid0 * 8 + 7) * 64) = (re, im) of frame fid0, (re, im) of frame fid0 + 1.  A wave evaluates that block for its next 64 pairs
// in one call (lane k = the pair of iteration k) and hands the values out by v_readlane -- round 4: as a third pass of the bin loop the one
// extra bin cost a whole wave-wide Philox call per frame, a third of the kernel's generator work.
template <int NFFT, int STFT_FPB>
__global__ __launch_bounds__(64 * STFT_FPB) void stft_kernel(
    const float* __restrict__ audio, int64_t audio_stride, int length, const float* __restrict__ window, int hop,
    int n_frames, int n_bins_out, int total_frames, float* __restrict__ y, float* __restrict__ x,
    const float* __restrict__ noise, float sigma, uint64_t seed, float* __restrict__ clip_absmax) {
  constexpr int NBUF = NFFT == 1024 ? 2 : 1;     // 512 / 256 points transform in place
  __shared__ float2 buf[NBUF][STFT_FPB][NFFT];
  __shared__ float2 tw[NFFT];                    // exp(-2 pi i q / N) for q < N (the radix-8 / radix-4 passes index up to 7 k N / (8 P) < N)
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int q = threadIdx.x; q < NFFT; q += blockDim.x) {
    float s, c;
    sincospif(-2.0f * (float)q / (float)NFFT, &s, &c);
    tw[q] = make_float2(c, s);
  }
  __syncthreads();
  const int64_t plane = (int64_t)n_frames * n_bins_out;
  const int npairs = (total_frames + 1) / 2;
  const int n_low = n_bins_out < NFFT / 2 ? n_bins_out : NFFT / 2;      // bins served by the paired blocks
  const bool last_bin = n_bins_out > NFFT / 2;
  const bool gen_last = last_bin && x != nullptr && noise == nullptr;
  const int pid_step = gridDim.x * STFT_FPB;
  float nyq[4] = {0.f, 0.f, 0.f, 0.f};
  int it = 0;
  for (int pid = blockIdx.x * STFT_FPB + wv; pid < npairs; pid += pid_step, ++it) {
    if (gen_last && (it & 63) == 0) {
      const int64_t pk = (int64_t)pid + (int64_t)lane * pid_step;       // the pair of iteration it + lane
      philox_normal4(seed, ((uint64_t)(2 * pk) * 8 + 7) * 64, nyq);
    }
    const int fid0 = 2 * pid, fid1 = fid0 + 1;
    const bool two_frames = fid1 < total_frames;
    const int b0 = fid0 / n_frames, t0 = fid0 % n_frames;
    const int b1 = two_frames ? fid1 / n_frames : b0, t1 = two_frames ? fid1 % n_frames : t0;
    const float* a0 = audio + (int64_t)b0 * audio_stride;
    const float* a1 = audio + (int64_t)b1 * audio_stride;
    for (int n = lane; n < NFFT; n += 64) {
      int j0 = t0 * hop + n - NFFT / 2, j1 = t1 * hop + n - NFFT / 2;
      if (j0 < 0) j0 = -j0;
      if (j0 >= length) j0 = 2 * (length - 1) - j0;
      if (j1 < 0) j1 = -j1;
      if (j1 >= length) j1 = 2 * (length - 1) - j1;
      const float wn = window[n];
      buf[0][wv][n] = make_float2(a0[j0] * wn, two_frames ? a1[j1] * wn : 0.f);
    }
    STFT_WAVE_SYNC();
    const int cur = fft_forward<NFFT>(&buf[0][wv][0], &buf[NBUF - 1][wv][0], tw, lane);
    // ---- separate the two spectra and write them (+ the noisy copies); `fr` = 0 / 1 selects the frame of the pair
#pragma unroll
    for (int fr = 0; fr < 2; ++fr) {
      if (fr == 1 && !two_frames) break;
      const int fid = fr ? fid1 : fid0, b = fr ? b1 : b0, t = fr ? t1 : t0;
      const int64_t row = ((int64_t)b * 2) * plane + (int64_t)t * n_bins_out;
      float* yre = y + row;
      float* yim = yre + plane;
      float amax = 0.f;
      auto bin = [&](int f) __attribute__((always_inline)) {
        const float2 z = buf[cur][wv][f], zn = buf[cur][wv][(NFFT - f) & (NFFT - 1)];
        return fr == 0 ? make_float2(0.5f * (z.x + zn.x), 0.5f * (z.y - zn.y)) : make_float2(0.5f * (z.y + zn.y), -0.5f * (z.x - zn.x));
      };
      // bins in pairs (f, f + 64): one Philox block = four normals = the noise of both
      for (int f0 = lane, jp = 0; f0 < n_low; f0 += 128, ++jp) {
        const int f1 = f0 + 64;
        const bool two = f1 < n_low;
        const float2 v0 = bin(f0), v1 = two ? bin(f1) : make_float2(0.f, 0.f);
        yre[f0] = v0.x;
        yim[f0] = v0.y;
        if (two) { yre[f1] = v1.x; yim[f1] = v1.y; }
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v1.x), fabsf(v1.y))));
        if (x != nullptr) {
          const int64_t o0 = row + f0;
          float g[4];
          if (noise != nullptr) {
            g[0] = noise[o0];
            g[1] = noise[o0 + plane];
            g[2] = two ? noise[o0 + 64] : 0.f;
            g[3] = two ? noise[o0 + 64 + plane] : 0.f;
          } else {
            philox_normal4(seed, ((uint64_t)fid * 8 + jp) * 64 + lane, g);
          }
          x[o0] = v0.x + sigma * g[0];
          x[o0 + plane] = v0.y + sigma * g[1];
          if (two) {
            x[o0 + 64] = v1.x + sigma * g[2];
            x[o0 + 64 + plane] = v1.y + sigma * g[3];
          }
        }
      }
      if (last_bin) {
        // the values of this pair sit in lane (it & 63) of the block evaluated above
        const float gre = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nyq[2 * fr]), it & 63));
        const float gim = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nyq[2 * fr + 1]), it & 63));
        if (lane == 0) {
          const float2 v = bin(NFFT / 2);
          const int64_t o = row + NFFT / 2;
          yre[NFFT / 2] = v.x;
          yim[NFFT / 2] = v.y;
          amax = fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y)));
          if (x != nullptr) {
            x[o] = v.x + sigma * (noise != nullptr ? noise[o] : gre);
            x[o + plane] = v.y + sigma * (noise != nullptr ? noise[o + plane] : gim);
          }
        }
      }
      if (clip_absmax != nullptr) {
        amax = wave_max(amax);
        if (lane == 0) atomicMax((unsigned int*)(clip_absmax + b), __float_as_uint(amax));  // amax >= 0
      }
    }
    STFT_WAVE_SYNC();      // the next pair overwrites buf[0][wv]
  }
}

// normalize_output_fft path (av_dataset.py:339-341): y *= 1/(max|y| + 1e-7) per clip, then x = y + sigma*N.
__global__ void stft_normalise_kernel(float* __restrict__ y, float* __restrict__ x, const float* __restrict__ noise,
                                      const float* __restrict__ clip_absmax, int64_t per_clip, int64_t total,
                                      int n_frames, int n_bins_out, float sigma, uint64_t seed) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / per_clip;
    const float s = 1.0f / (clip_absmax[b] + 1e-7f);
    const float v = y[i] * s;
    y[i] = v;
    if (x != nullptr) {
      float nz;
      if (noise != nullptr) {
        nz = noise[i];
      } else {
        const int64_t r = i - b * per_clip;
        const int64_t plane = (int64_t)n_frames * n_bins_out;
        const int pl = (int)(r / plane);
        const int64_t tf = r - pl * plane;  // t * n_bins + f
        float g[4];
        philox_normal4(seed, (uint64_t)(b * n_frames) * n_bins_out + tf, g);
        nz = g[pl];
      }
      x[i] = v + sigma * nz;
    }
  }
}

extern "C" int maavss_stft_fwd(const float* audio, int64_t batch, int64_t length, int64_t audio_stride,
                               const float* window, int n_fft, int hop, int n_frames, int n_bins_out, float* y,
                               float* x, const float* noise, float sigma, uint64_t seed, float* clip_absmax,
                               void* stream) {
  MAAVSS_CHECK_ARG(n_fft == 256 || n_fft == 512 || n_fft == 1024, "stft: n_fft must be 256, 512 or 1024 (got %d)", n_fft);
  MAAVSS_CHECK_ARG(audio && window && y, "stft: null pointer");
  MAAVSS_CHECK_ARG(batch > 0 && hop > 0 && n_frames > 0, "stft: empty problem");
  MAAVSS_CHECK_ARG(n_bins_out >= 1 && n_bins_out <= n_fft / 2 + 1, "stft: n_bins_out out of range");
  MAAVSS_CHECK_ARG(length > n_fft / 2, "stft: reflect padding needs length > n_fft/2");
  MAAVSS_CHECK_ARG((int64_t)(n_frames - 1) * hop + n_fft / 2 - 1 < 2 * length - 1, "stft: frames run past the reflected signal");
  const int total = (int)(batch * n_frames);
  hipStream_t st = (hipStream_t)stream;
  // grid-stride over the frames: at most 8 workgroups per CU worth of blocks (the twiddle table is built once per workgroup)
#define LAUNCH(N, FPB)                                                                                         \
  hipLaunchKernelGGL((stft_kernel<N, FPB>), dim3(cdiv(cdiv(total, 2), FPB) < 2048 ? cdiv(cdiv(total, 2), FPB) : 2048), dim3(64 * FPB), 0, st, audio, audio_stride, \
                     (int)length, window, hop, n_frames, n_bins_out, total, y, x, noise, sigma, seed, clip_absmax)
  if (n_fft == 256) LAUNCH(256, 4);
  else if (n_fft == 512) LAUNCH(512, 4);
  else LAUNCH(1024, 2);
#undef LAUNCH
  MAAVSS_LAUNCH_CHECK("stft_kernel");
  return MAAVSS_OK;
}

extern "C" int maavss_stft_normalise(float* y, float* x, const float* noise, const float* clip_absmax, int64_t batch,
                                     int n_frames, int n_bins_out, float sigma, uint64_t seed, void* stream) {
  MAAVSS_CHECK_ARG(y && clip_absmax, "stft_normalise: null pointer");
  const int64_t per_clip = 2LL * n_frames * n_bins_out, total = per_clip * batch;
  int grid = cdiv(total, 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(stft_normalise_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, x, noise, clip_absmax,
                     per_clip, total, n_frames, n_bins_out, sigma, seed);
  MAAVSS_LAUNCH_CHECK("stft_normalise_kernel");
  return MAAVSS_OK;
}

// ---- inverse STFT (AV_Dataset.istft, av_dataset.py:181-201: torch.istft(n_fft, hop, win_length = n_fft, window,
// normalized, onesided, center)) -- the audio side of SURVEY.md 8 row f2 (mask / autoencoder output -> waveform).
// Pass 1, one wavefront per frame: the one-sided bins of the [B,2,T,F] tensor are mirrored into the Hermitian
// spectrum in LDS (imaginary parts of DC and Nyquist dropped as irfft does; a trimmed Nyquist bin reads as zero, the
// reference pads it), the same Stockham FFT runs with conjugated twiddles, and the real part, scaled by
// (normalized ? sqrt(N) : 1) / N and multiplied by the synthesis window, goes to frames[B][T][N].
// Pass 2, one thread per output sample: overlap-add of the <= ceil(N / hop) frames that cover it, divided by the
// window-square envelope, with the N/2 samples of centre padding cut off (length hop * (T - 1)).
template <int NFFT, int FPB>
__global__ __launch_bounds__(64 * FPB) void istft_frames_kernel(const float* __restrict__ spec, const float* __restrict__ window,
                                                                int n_frames, int n_bins_in, int total_frames, float scale,
                                                                float* __restrict__ frames) {
  constexpr int LOG2N = NFFT == 256 ? 8 : (NFFT == 512 ? 9 : 10);
  __shared__ float2 buf[2][FPB][NFFT];
  __shared__ float2 tw[NFFT / 2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int q = threadIdx.x; q < NFFT / 2; q += blockDim.x) {
    float s, c;
    sincospif(2.0f * (float)q / (float)NFFT, &s, &c);   // conjugate twiddles: inverse transform
    tw[q] = make_float2(c, s);
  }
  const int fid = blockIdx.x * FPB + wv;
  const bool active = fid < total_frames;
  const int b = active ? fid / n_frames : 0, t = active ? fid % n_frames : 0;
  const int64_t plane = (int64_t)n_frames * n_bins_in;
  const float* re = spec + ((int64_t)b * 2) * plane + (int64_t)t * n_bins_in;
  const float* im = re + plane;
  for (int k = lane; k <= NFFT / 2; k += 64) {
    float2 v = make_float2(0.f, 0.f);
    if (active && k < n_bins_in) v = make_float2(re[k], (k == 0 || k == NFFT / 2) ? 0.f : im[k]);
    buf[0][wv][k] = v;
    if (k > 0 && k < NFFT / 2) buf[0][wv][NFFT - k] = make_float2(v.x, -v.y);
  }
  __syncthreads();
  int cur = 0;
#pragma unroll
  for (int s = 0; s < LOG2N; ++s) {
    const int p = 1 << s;
    for (int i = lane; i < NFFT / 2; i += 64) {
      const int k = i & (p - 1);
      float2 u0 = buf[cur][wv][i];
      float2 u1 = buf[cur][wv][i + NFFT / 2];
      float2 w = tw[k * (NFFT / (2 * p))];
      float2 v = make_float2(u1.x * w.x - u1.y * w.y, u1.x * w.y + u1.y * w.x);
      const int j = ((i - k) << 1) + k;
      buf[cur ^ 1][wv][j] = make_float2(u0.x + v.x, u0.y + v.y);
      buf[cur ^ 1][wv][j + p] = make_float2(u0.x - v.x, u0.y - v.y);
    }
    __syncthreads();
    cur ^= 1;
  }
  if (!active) return;
  float* out = frames + (int64_t)fid * NFFT;
  for (int n = lane; n < NFFT; n += 64) out[n] = buf[cur][wv][n].x * scale * window[n];
}

__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, const float* __restrict__ window,
                                                        int n_fft, int hop, int n_frames, int out_len, int64_t out_stride,
                                                        int64_t total, float* __restrict__ audio) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int b = (int)(i / out_len), n = (int)(i % out_len);
    const int pos = n + n_fft / 2;                       // position in the centre-padded signal
    int t_hi = pos / hop;
    if (t_hi > n_frames - 1) t_hi = n_frames - 1;
    int t_lo = (pos - n_fft + hop) / hop;                // smallest t with t*hop + n_fft > pos
    if (pos - n_fft + 1 <= 0) t_lo = 0;
    float num = 0.f, den = 0.f;
    for (int t = t_lo; t <= t_hi; ++t) {
      const int j = pos - t * hop;
      if (j < 0 || j >= n_fft) continue;
      num += frames[((int64_t)b * n_frames + t) * n_fft + j];
      den += window[j] * window[j];
    }
    audio[(int64_t)b * out_stride + n] = num / den;
  }
}

extern "C" int maavss_istft(const float* spec, int64_t batch, int n_frames, int n_bins_in, const float* window, int n_fft,
                            int hop, int normalized, float* frames_ws, float* audio, int64_t audio_stride, void* stream) {
  MAAVSS_CHECK_ARG(n_fft == 256 || n_fft == 512 || n_fft == 1024, "istft: n_fft must be 256, 512 or 1024 (got %d)", n_fft);
  MAAVSS_CHECK_ARG(spec && window && frames_ws && audio, "istft: null pointer");
  MAAVSS_CHECK_ARG(batch > 0 && hop > 0 && n_frames > 1, "istft: needs at least two frames");
  MAAVSS_CHECK_ARG(n_bins_in == n_fft / 2 || n_bins_in == n_fft / 2 + 1, "istft: n_bins must be n_fft/2 (trimmed) or n_fft/2+1");
  MAAVSS_CHECK_ARG(hop <= n_fft, "istft: hop larger than the window leaves gaps (window envelope would be zero)");
  const int out_len = hop * (n_frames - 1);
  MAAVSS_CHECK_ARG(audio_stride >= out_len, "istft: audio_stride smaller than hop*(n_frames-1)");
  MAAVSS_CHECK_ARG((int64_t)(n_frames - 1) * hop + n_fft >= out_len + n_fft / 2, "istft: frames do not cover the output");
  const int total = (int)(batch * n_frames);
  const float scale = (normalized ? sqrtf((float)n_fft) : 1.f) / (float)n_fft;
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH(N, FPB)                                                                                                   \
  hipLaunchKernelGGL((istft_frames_kernel<N, FPB>), dim3(cdiv(total, FPB)), dim3(64 * FPB), 0, st, spec, window, n_frames, \
                     n_bins_in, total, scale, frames_ws)
  if (n_fft == 256) LAUNCH(256, 4);
  else if (n_fft == 512) LAUNCH(512, 4);
  else LAUNCH(1024, 2);
#undef LAUNCH
  MAAVSS_LAUNCH_CHECK("istft_frames_kernel");
  const int64_t n_out = batch * out_len;
  int grid = cdiv(n_out, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(istft_ola_kernel, dim3(grid), dim3(256), 0, st, frames_ws, window, n_fft, hop, n_frames, out_len, audio_stride,
                     n_out, audio);
  MAAVSS_LAUNCH_CHECK("istft_ola_kernel");
  return MAAVSS_OK;
}
