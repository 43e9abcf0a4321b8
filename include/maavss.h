/* libmaavss_hip -- C-ABI of the MI355X (gfx950) kernels behind the MAAVSS training hot path.
 *
 * The reference (carlmoore256/MAAVSS) has no FFI layer: its boundary is two Python classes
 * (avse_model_final.py:14 AV_Fusion_Model_Frames, video_attention.py:24 VideoAttention) plus the
 * STFT helpers of av_dataset.py.  Each entry point below replaces the ATen/vendor-library work one
 * reference line dispatches; the citation names that line.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - plain pointers are DEVICE pointers unless named host_*; the library never allocates, frees
 *     or synchronises; every launch goes to `stream` (a hipStream_t, 0 = default stream);
 *   - scratch memory is caller-provided (`ws`, size documented per call);
 *   - return value 0 = ok, non-zero = error, text via maavss_last_error() (thread-local);
 *   - tensors are dense, row-major in the documented order; f32 unless stated; "bf16" = raw
 *     uint16 bfloat16 bits;
 *   - `precise` selects the arithmetic of MFMA-backed kernels: 0 = operands rounded to bf16,
 *     f32 accumulate (v_mfma_f32_16x16x32_bf16); 1 = exact f32 MFMA (v_mfma_f32_16x16x4_f32).
 */
#ifndef MAAVSS_H
#define MAAVSS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* maavss_last_error(void);
int maavss_version(void);
const char* maavss_arch(void);

/* ---- K17 audio STFT + noise ------------------------------------------------------------------
 * replaces AV_Dataset.stft (av_dataset.py:157-174: torchaudio spectrogram(pad=0, hamming, n_fft,
 * hop, power=None, normalized, onesided) minus the last frame [and last bin]) and
 * gen_stft_example/add_noise (av_dataset.py:217-220, 335-342).
 * audio [batch][audio_stride] (length valid samples), window [n_fft] = periodic Hamming already
 * multiplied by the 1/sqrt(sum w^2) normalisation, y/x [batch][2][n_frames][n_bins_out].
 * x (nullable) = y + sigma * noise; noise (nullable, same layout as y) else Philox4x32-10(seed).
 * clip_absmax (nullable) [batch], pre-zeroed: receives max|y| per clip (for maavss_stft_normalise). */
int maavss_stft_fwd(const float* audio, int64_t batch, int64_t length, int64_t audio_stride, const float* window,
                    int n_fft, int hop, int n_frames, int n_bins_out, float* y, float* x, const float* noise,
                    float sigma, uint64_t seed, float* clip_absmax, void* stream);
/* normalize_output_fft=True path (av_dataset.py:339-341): y *= 1/(clip_absmax + 1e-7); x = y + sigma*noise. */
int maavss_stft_normalise(float* y, float* x, const float* noise, const float* clip_absmax, int64_t batch,
                          int n_frames, int n_bins_out, float sigma, uint64_t seed, void* stream);

#ifdef __cplusplus
}
#endif
#endif
