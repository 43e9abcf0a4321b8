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
 *   - `precise` (a.k.a. mode) selects the arithmetic of MFMA-backed kernels: 0 = operands rounded to bf16,
 *     f32 accumulate (v_mfma_f32_16x16x32_bf16); 1 = exact f32 MFMA (v_mfma_f32_16x16x4_f32); 2 = operands
 *     rounded to IEEE half, f32 accumulate (v_mfma_f32_16x16x32_f16; same rate as bf16, used for forward
 *     operands whose range BatchNorm bounds).
 */
#ifndef MAAVSS_H
#define MAAVSS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ABI version = what maavss_version() of a matching library returns; bumped whenever an entry point is removed or changes
 * meaning (INTEGRATION.md "ABI history").  maavss_amd/_lib.py refuses a library whose version differs from this header's.
 *   100  rounds 1-2
 *   300  round 3: maavss_vit_attn_fp8{,_ws_bytes} removed (-> maavss_vit_attn_mx*); maavss_conv3d_c1_fwd(precise = 2) writes one
 *        stat_partials row per 8-tile workgroup (maavss_conv3d_c1_fwd_nparts), not one per tile; the Philox counter layout of
 *        maavss_stft_fwd's in-kernel noise changed (same seed, different noise)
 *   400  round 4: maavss_set_deterministic_workspace takes the stream the scratch is bound to; NULL ln_gamma / ln_beta = LayerNorm without
 *        the affine part; epilogue 4 of maavss_vit_ws_gemm (GELU in packed half); the in-kernel noise of maavss_stft_fwd draws the bin n_fft / 2
 *        from its own per-frame-pair Philox block (same seed, different noise in that bin); maavss_bn_pool_act_fwd and
 *        maavss_conv3d_c1_bn_pool_act take an out_bf16 pointer behind out16, maavss_conv3d_wgrad's dy16 became the mask in16; see INTEGRATION.md */
#define MAAVSS_ABI_VERSION 400
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

/* inverse: AV_Dataset.istft (av_dataset.py:181-201) = torch.istft(n_fft, hop, win_length = n_fft, window, normalized,
 * onesided, center).  spec [batch][2][n_frames][n_bins_in] (n_bins_in = n_fft/2+1, or n_fft/2 when the Nyquist bin was
 * trimmed: read as zero), window [n_fft] = the RAW periodic Hamming window; frames_ws [batch][n_frames][n_fft] scratch;
 * audio [batch][audio_stride], hop*(n_frames-1) valid samples per clip. */
int maavss_istft(const float* spec, int64_t batch, int n_frames, int n_bins_in, const float* window, int n_fft, int hop,
                 int normalized, float* frames_ws, float* audio, int64_t audio_stride, void* stream);

/* ---- generic f32 GEMM on MFMA -----------------------------------------------------------------
 * C[M,N] = act(alpha * op(A)[M,K] . op(B)[N,K]^T) (+ C when beta = 1).  Replaces the nn.Linear forwards
 * of avse_model_final.py:141-146,203-213,244-249,264-268, the LSTM input projection (:132,242) and the
 * gradients autograd derives from them.  transA=0: A[m*lda+k], 1: A[k*lda+m]; transB=0: B[n*ldb+k]
 * (torch Linear weight layout), 1: B[k*ldb+n]; transC=0: C[m*ldc+n], 1: C[n*ldc+m].
 * act: 0 none, 1 tanh, 2 sigmoid.  split_k: 0 = auto, n>1 = split K over n blocks (f32 atomics). */
/* Process-wide switch (default 0).  1: maavss_gemm_f32 takes no path that accumulates with f32 atomics (the split-K forms of the
 * M = batch Linear layers and the automatic split-K of the generic kernel) -- bit-identical results run to run, at the price of
 * slower weight streaming on those shapes.  Everything else in the library is deterministic already (conv weight gradients:
 * partial + reduce).  No reference counterpart (torch on one device is deterministic for these layers). */
int maavss_set_deterministic(int on);   /* returns the previous setting */
int maavss_get_deterministic(void);
/* Optional device scratch (16-byte aligned; the library never allocates) that lets deterministic mode keep the K split of the
 * M = batch Linear forms: slices write partial sums [slices][32][cols], a second kernel adds them in slice order.  8.4 MB covers the
 * reference shapes (fc1 8192 -> 4096 at 512-wide slices); shapes that need more fall back to one slice per output element.  One
 * scratch per process, bound to `stream`: Linear kernels launched on any other stream do not use it (single-slice path), so
 * two streams cannot race on it (ABI 400; before: no stream argument and no guard).  NULL removes it. */
int maavss_set_deterministic_workspace(float* ws, int64_t bytes, void* stream);
int maavss_gemm_f32(const float* A, int64_t lda, int transA, const float* B, int64_t ldb, int transB, float* C,
                    int64_t ldc, int transC, int64_t M, int64_t N, int64_t K, float alpha, int beta, int act,
                    int split_k, int precise, void* stream);

/* ---- K7/K8 Conv3d(k=(3,5,5), stride 1, pad (1,p,p), bias=False) -- avse_model_final.py:34,39,44,49,54 ----
 * Activations channels-last [B][T][H][W][C] f32.  Weights in the reference layout [Co][Ci][3][5][5].
 * maavss_conv3d_kp(c_in): padded K of the re-laid weight image; prep writes wt[3][n][KP] in bf16 (precise=0)
 * or f32 (precise=1): mode 0 = forward (n=Co), mode 1 = input gradient (n=Ci; flipped taps, use pad 4-p).
 * igemm: y[B][T][Ho][Wo][c_out], Ho = H+2*pad-4; stat_partials (nullable) [grid blocks][2][c_out] receives
 * per-block (sum, sum^2) for BatchNorm, grid blocks = ceil(Wo/16)*ceil(Ho/16)*B*T.
 * wgrad: dw[Co][Ci][3][5][5] (+= when beta=1); ws of maavss_conv3d_wgrad_ws_bytes(c_in,c_out,nchunk) bytes.
 * x16 / dy16 = 1: that operand is already stored in the MFMA format of `precise` (IEEE half x from
 * maavss_bn_pool_act_fwd's out16 for the forward pass, bf16 dy from maavss_bn_pool_act_bwd for the two backward passes):
 * the producer rounded once, the kernels copy instead of converting -- bit-identical results, half the bytes. */
int maavss_conv3d_kp(int c_in);
int maavss_conv3d_prep_weights(const float* w, void* wt, int c_out, int c_in, int mode, int precise, void* stream);
int maavss_conv3d_igemm(const void* x, const void* wt, float* y, float* stat_partials, int B, int T, int H, int W,
                        int c_in, int c_out, int pad, int precise, int x16, void* stream);
int64_t maavss_conv3d_wgrad_ws_bytes(int c_in, int c_out, int nchunk);
/* in16: bit 0 = dy is bf16 (precise = 0), bit 1 = x is bf16 as well (needs bit 0 and one of the shapes 16->32, 32->64, 64->64: both operand
 * images then go global -> LDS by DMA, no conversion); any combination gives the same dw bit for bit. */
int maavss_conv3d_wgrad(const void* x, const void* dy, float* dw, float* ws, int nchunk, int B, int T, int H, int W,
                        int c_in, int c_out, int pad, int beta, int precise, int in16, void* stream);
/* first layer (C_in = 1, pad 2): x [B][T][H][W], w [16][1][3][5][5], w16_ws 1200 floats scratch,
 * y [B][T][H][W][16]; stat_partials [maavss_conv3d_c1_fwd_nparts(...)][2][16] (one row per workgroup: the MFMA form walks
 * 8 tiles per workgroup); wgrad ws = nchunk*1200 floats. */
int64_t maavss_conv3d_c1_fwd_nparts(int B, int T, int H, int W, int precise);
int maavss_conv3d_c1_fwd(const float* x, const float* w, float* w16_ws, float* y, float* stat_partials, int B, int T,
                         int H, int W, int precise /* 1: exact-f32 VALU convolution; 2: IEEE-half operands on the MFMA
                         (25 taps of a kd plane = one 32-deep step), f32 accumulation */, void* stream);
int maavss_conv3d_c1_wgrad(const float* x, const float* dy, float* dw, float* ws, int nchunk, int B, int T, int H,
                           int W, int beta, void* stream);
/* The same with the first layer's BatchNorm / max-pool / LeakyReLU backward folded into the loader: y is the conv output,
 * dout / out / argmax [B*T][H/pool][W/pool][16] the pooled gradient, output and window argmax, coef [3][16] the
 * coefficients maavss_bn_pool_act_bwd leaves at ws + 2*C*maavss_bn_stats_nblk(rows) when called with dy = NULL. */
int maavss_conv3d_c1_wgrad_bn(const float* x, const float* y, const float* dout, const float* out, const void* argmax,
                              const float* mean, const float* invstd, const float* coef, int pool, float* dw, float* ws,
                              int nchunk, int B, int T, int H, int W, int beta,
                              int precise /* 1: exact-f32 VALU; 0: bf16 operands on the MFMA (positions = K dimension) */, void* stream);

/* The 16-bit first layer WITHOUT its conv output (avse_model_final.py:34-37: Conv3d(1,16,(3,5,5)) -> BatchNorm3d -> MaxPool3d((1,2,2))
 * -> LeakyReLU): y [B][T][H][W][16] is the largest tensor of the step (1.6 GB at 32 x 16 x 224^2) and the layer is 59 GFLOP on
 * an idle matrix pipe, so the convolution is run three times instead of stored once and read twice:
 *   maavss_conv3d_c1_stats          conv (IEEE-half MFMA) -> stat_partials [maavss_conv3d_c1_fwd_nparts(.., 2)][2][16] only; `y` is
 *                                   written ONLY if some |gamma[c]| < 1e-2 (the BatchNorm backward reduction then gathers from it);
 *   maavss_conv3d_c1_bn_pool_act    conv again -> gamma (y - mean) invstd + beta -> 2x2 max pool -> LeakyReLU(0.01): out f32 and out16
 *                                   (IEEE half, may be NULL) [B*T][H/2][W/2][16], argmax one byte per element (window position
 *                                   dy * 2 + dx); bit-identical to maavss_conv3d_c1_fwd(.., 2) + maavss_bn_pool_act_fwd;
 *   maavss_conv3d_c1_wgrad_bn_recompute   maavss_conv3d_c1_wgrad_bn(.., precise 0) with w [16][1][3][5][5] in place of y and the
 *                                   BatchNorm bias bn_beta [16] in place of the pooled output: the tile's y is recomputed from the
 *                                   staged halo and the LeakyReLU slope from its sign; bit-identical gradient. */
int maavss_conv3d_c1_stats(const float* x, const float* w, const float* gamma, float* y, float* stat_partials, int B, int T, int H,
                           int W, void* stream);
int maavss_conv3d_c1_bn_pool_act(const float* x, const float* w, const float* mean, const float* invstd, const float* gamma,
                                 const float* beta, float* out, void* out16, void* out_bf16 /* nullable, as maavss_bn_pool_act_fwd's */,
                                 void* argmax, int B, int T, int H, int W, void* stream);
int maavss_conv3d_c1_wgrad_bn_recompute(const float* x, const float* w, const float* dout, const void* argmax, const float* mean,
                                        const float* invstd, const float* bn_beta, const float* coef, int pool, float* dw, float* ws,
                                        int nchunk, int B, int T, int H, int W, int beta, void* stream);

/* ---- K9 BatchNorm (train mode) + MaxPool(1,p,p) + LeakyReLU / BatchNorm2d + Tanh -----------------
 * avse_model_final.py:35-37,...,55-57 (pool before activation) and :103-104 (pool = 1, act = 1).
 * y channels-last [B*T][H][W][C]; the pooled output (and its gradient) are addressed with element strides
 * b*os_b + t*os_t + (py*Wp+px)*os_p + c*os_c so the last stage can write the [B,16,T,S] / LSTM-sequence
 * layout directly (avse_model_final.py:58,239-240).  act: 0 LeakyReLU(0.01), 1 Tanh.
 * bn_finalize: count = B*T*H*W; updates running stats with `momentum` and the unbiased variance and
 * increments num_batches_tracked (int64) like torch.nn.BatchNorm (pointers nullable).
 * bwd ws: (2*C*nblk + 3*C) floats, nblk = maavss_bn_stats_nblk(B*T*Hp*Wp). */
int maavss_bn_stats_nblk(int64_t rows);
int maavss_bn_stats(const float* y, float* partials, int64_t rows, int C, void* stream);
int maavss_bn_finalize(const float* partials, int nblk, int C, double count, float eps, float momentum, float* mean,
                       float* invstd, float* running_mean, float* running_var, void* num_batches_tracked,
                       float* ws /* nullable; 256*2*C floats enable the two-level reduction */, void* stream);
/* eval mode (model.eval()): mean / invstd from the running statistics instead of the batch */
int maavss_bn_eval_stats(const float* running_mean, const float* running_var, float eps, float* mean, float* invstd, int C,
                         void* stream);
int maavss_bn_pool_act_fwd(const float* y, const float* mean, const float* invstd, const float* gamma,
                           const float* beta, float* out, void* argmax, int B, int T, int H, int W, int C, int pool,
                           int act, int64_t os_b, int64_t os_t, int64_t os_p, int64_t os_c,
                           void* out16 /* nullable: IEEE-half copy of the pooled activation, contiguous channels-last
                                          [B][T][Hp][Wp][C] -- the operand format of the next Conv3d's forward MFMA */,
                           void* out_bf16 /* nullable: the same as bf16 -- the x operand of the next Conv3d's weight gradient
                                             (maavss_conv3d_wgrad, in16 bit 1) */,
                           void* stream);
/* beta (nullable): with it, LeakyReLU layers recover the normalised input at the pooled maximum from `out` instead of
 * gathering it from y (xhat = (leaky^-1(out) - beta) / gamma; channels with |gamma| < 1e-2 still gather).  dy NULL: only
 * dgamma / dbeta and the coefficients (see maavss_conv3d_c1_wgrad_bn). */
int maavss_bn_pool_act_bwd(const float* dout, const float* out, const void* argmax, const float* y,
                           const float* mean, const float* invstd, const float* gamma, const float* beta, void* dy, float* dgamma,
                           float* dbeta, int accumulate, float* ws, int B, int T, int H, int W, int C, int pool,
                           int act, int64_t os_b, int64_t os_t, int64_t os_p, int64_t os_c,
                           int dy_bf16 /* 1: dy is written as bf16 (what both of its MFMA consumers round it to), 0: f32 */,
                           void* stream);

/* Global-batch BatchNorm under data parallelism (the reference normalises over the WHOLE batch on one device,
 * avse_model_final.py:35,40,45,50,55,103): split forms whose per-channel sums -- double sums[2*C + 1] = {sum, sum^2 (or
 * sum g, sum g*z), element count} -- are all-reduced by the host (RCCL) between the two halves.  bwd_finish takes this
 * rank's sums for dgamma / dbeta (the gradient all-reduce adds the ranks) and the reduced ones for the dx coefficients;
 * coef: 3*C floats (as left by maavss_bn_pool_act_bwd in its ws). */
int maavss_bn_partials_to_sums(const float* partials, int nblk, int C, double count, double* sums,
                               float* ws /* nullable, as bn_finalize */, void* stream);
int maavss_bn_finalize_sums(const double* sums, int C, float eps, float momentum, float* mean, float* invstd,
                            float* running_mean, float* running_var, void* num_batches_tracked, void* stream);
int maavss_bn_pool_act_bwd_sums(const float* dout, const float* out, const void* argmax, const float* y, const float* mean,
                                const float* invstd, const float* gamma, const float* beta, float* ws, double* sums, int B,
                                int T, int H, int W, int C, int pool, int act, int64_t os_b, int64_t os_t, int64_t os_p,
                                int64_t os_c, void* stream);
int maavss_bn_pool_act_bwd_finish(const float* dout, const float* out, const void* argmax, const float* y, const float* mean,
                                  const float* invstd, const float* gamma, const float* beta, void* dy, float* dgamma,
                                  float* dbeta, int accumulate, const double* sums_local, const double* sums_global,
                                  float* coef, int B, int T, int H, int W, int C, int pool, int act, int64_t os_b,
                                  int64_t os_t, int64_t os_p, int64_t os_c, int dy_bf16, void* stream);

/* ---- K10 Conv2d(k=(3,9), stride (sh,sw), pad (1,pw), bias=False) -- avse_model_final.py:98-102 ----
 * in_layout 0: x NCHW [B][Ci][H][W] (network input), 1: NHWC; y/dy NHWC [B][Ho][Wo][Co]; w [Co][Ci][3][9].
 * wgrad ws: maavss_conv2d_wgrad_nchunk(B,Ho,Wo,Ci,Co) * Co*Ci*27 floats. */
int maavss_conv2d_fwd(const float* x, const float* w, float* y, int B, int Ci, int H, int W, int Co, int sh, int sw,
                      int pw, int in_layout, void* stream);
int maavss_conv2d_dgrad(const float* dy, const float* w, float* dx, int B, int Ci, int H, int W, int Co, int sh,
                        int sw, int pw, void* stream);
int maavss_conv2d_wgrad_nchunk(int B, int Ho, int Wo, int Ci, int Co);
int maavss_conv2d_wgrad(const float* x, const float* dy, float* dw, float* ws, int B, int Ci, int H, int W, int Co,
                        int sh, int sw, int pw, int in_layout, int beta, void* stream);

/* ---- K11 ConvTranspose2d(k=(3,kw), kw in {9,10}, stride (sh,sw), padding (1,4), output_padding (oph,opw), bias=False)
 * -- the STFT decoder, avse_model_final.py:155-193 (audio_ae_forward :254-256) ----
 * x / dx NHWC [B][Hi][Wi][Ci]; w [Ci][Co][3][kw] (reference layout); y / dy [B][Ho][Wo][Co] for out_layout 1 or the
 * network's NCHW [B][Co][Ho][Wo] for out_layout 0 (last decoder layer); Ho = (Hi-1) sh + 1 + oph, Wo = (Wi-1) sw - 8 + kw + opw.
 * wgrad ws: maavss_convt2d_wgrad_nchunk(B,Hi,Wi) * Ci*Co*3*kw floats; beta 1 accumulates into dw. */
int maavss_convt2d_out_size(int Hi, int Wi, int kw, int sh, int sw, int oph, int opw, int* Ho, int* Wo);
int maavss_convt2d_fwd(const float* x, const float* w, float* y, int B, int Ci, int Hi, int Wi, int Co, int kw, int sh, int sw,
                       int oph, int opw, int out_layout, void* stream);
int maavss_convt2d_dgrad(const float* dy, const float* w, float* dx, int B, int Ci, int Hi, int Wi, int Co, int kw, int sh,
                         int sw, int oph, int opw, int out_layout, void* stream);
int maavss_convt2d_wgrad_nchunk(int B, int Hi, int Wi);
int maavss_convt2d_wgrad(const float* x, const float* dy, float* dw, float* ws, int B, int Ci, int Hi, int Wi, int Co, int kw,
                         int sh, int sw, int oph, int opw, int out_layout, int beta, void* stream);

/* ---- K19 generic biased Conv2d / ConvTranspose2d of the phasegram variant avse_model.AV_Fusion_Model
 * (avse_model.py:433,452,494,591): kernels (1,9) and (5,5), any stride / padding, <= 25 taps.
 * A small map S [B][Hs][Ws][Cs] and a big map G [B][Hb][Wb][Cb] with by = sy*sh - ph + kh, bx = sx*sw - pw + kw and a
 * weight w[cs][cb][kh][kw] (= Conv2d's [Co][Ci] and ConvTranspose2d's [Ci][Co]):
 *   gen_small: S = bias + G (*) w   -- Conv2d forward, ConvTranspose2d input gradient (bias NULL)
 *   gen_big:   G = bias + S (*)^T w -- ConvTranspose2d forward, Conv2d input gradient (bias NULL)
 *   gen_wgrad: dw = S (x) G         -- both; ws: maavss_conv2d_gen_wgrad_nchunk(B,Hs,Ws) * Cs*Cb*kh*kw floats
 * small_strides / big_strides: HOST arrays of 4 element strides {batch, y, x, channel} (NCHW and channels-last, also
 * channel-padded, without copies).  channel_sum: out[c] (+)= sum_rows x[row*row_stride + c*chan_stride] (bias gradients). */
int maavss_conv2d_gen_small(const float* big, const float* w, const float* bias, float* small, int B, int Cs, int Hs, int Ws,
                            int Cb, int Hb, int Wb, int kh, int kw, int sh, int sw, int ph, int pw,
                            const int64_t* small_strides, const int64_t* big_strides, void* stream);
int maavss_conv2d_gen_big(const float* small, const float* w, const float* bias, float* big, int B, int Cs, int Hs, int Ws,
                          int Cb, int Hb, int Wb, int kh, int kw, int sh, int sw, int ph, int pw,
                          const int64_t* small_strides, const int64_t* big_strides, void* stream);
int maavss_conv2d_gen_wgrad_nchunk(int B, int Hs, int Ws);
int maavss_conv2d_gen_wgrad(const float* small, const float* big, float* dw, float* ws, int B, int Cs, int Hs, int Ws, int Cb,
                            int Hb, int Wb, int kh, int kw, int sh, int sw, int ph, int pw, const int64_t* small_strides,
                            const int64_t* big_strides, int beta, void* stream);
int maavss_channel_sum(const float* x, float* out, int64_t rows, int C, int64_t row_stride, int64_t chan_stride, int beta,
                       void* stream);
/* Linear bias + LeakyReLU(slope) (avse_model.py:613-621,660-663): z = act(z + bias) in place over [rows][n], act 0 = none,
 * 3 = LeakyReLU; leaky_bwd: dz = dout * (out > 0 ? 1 : slope). */
int maavss_bias_act_fwd(float* z, const float* bias, int64_t rows, int n, int act, float slope, void* stream);
int maavss_leaky_bwd(const float* dout, const float* out, float* dz, int64_t n, float slope, void* stream);

/* ---- K20 utilities.video_phasegram (utilities.py:206-228; train_av_net.py:122-125) ----
 * frames [batch][T][P][P] (attention maps, P in {32, 64}); out [batch][T][P*P] (= the reference's [B,1,T,P*P]):
 * fft2 -> fftshift over ALL axes, batch and frame included (the reference passes no dim) -> angle -> (cumulative ? cumsum / (2 pi P*P) : (angle + pi) / 2 pi) -> (diff ? temporal difference with a
 * zero first row) -> (normalize ? / max |.| over the whole batch tensor).  p_ws [batch][T][P*P] and absmax_ws [1] are scratch. */
int maavss_video_phasegram(const float* frames, int64_t batch, int T, int P, int diff, int cumulative, int normalize, float* p_ws,
                           float* absmax_ws, float* out, void* stream);

/* bilinear resize in front of the phasegram (utilities.py:208-209: torchvision resize of a tensor =
 * torch.nn.functional.interpolate(mode="bilinear", align_corners=False), no antialias): in [n][H][W] -> out [n][h][w]. */
int maavss_resize_bilinear(const float* in, float* out, int64_t n, int H, int W, int h, int w, void* stream);

/* ---- K12 bidirectional LSTM recurrence (hidden 256, no bias) -- avse_model_final.py:132-133,242 ----
 * gx [B][L][2][4][256] = X.W_ih^T (both directions, gate order i,f,g,o); av [B][L][512]; hp [B][L][2][256];
 * gs [B][L][2][4][256]; cs [B][L][2][256]; bwd: dav [B][L][512] -> dgx (same shape as gx), dc scratch [2][B][256]. */
int maavss_lstm_fwd(const float* gx, const float* whh_f, const float* whh_b, float* av, float* hp, float* gs,
                    float* cs, int B, int L, void* stream);
int maavss_lstm_bwd(const float* dav, const float* whh_f, const float* whh_b, const float* gs, const float* cs,
                    float* dgx, float* dc, int B, int L, void* stream);

/* ---- K15/K16 loss, activation backward, Adam ---------------------------------------------------
 * act_bwd: dz = dout * act'(out), act 1 tanh / 2 sigmoid (avse_model_final.py:246-249,264-268).
 * mse_pair: losses[0..2] = a_loss, v_loss, (a_loss + coeff*v_loss)*inv_num_seq and (nullable) the gradients
 * of losses[2] w.r.t. the predictions (train_avse_frames.py:166-170); ws = 1024 floats.
 * adam_step: torch.optim.Adam semantics (train_avse_frames.py:92,180) on flat 16-byte-aligned buffers;
 * `step` is the 1-based step count, grad_scale multiplies g (1/world_size for data parallel). */
int maavss_act_bwd(const float* dout, const float* out, float* dz, int64_t n, int act, void* stream);
int maavss_mse_pair(const float* a_pred, const float* a_tgt, int64_t na, const float* v_pred, const float* v_tgt,
                    int64_t nv, float coeff, float inv_num_seq, float* d_a, float* d_v, float* losses, float* ws,
                    void* stream);
int maavss_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                     float eps, int64_t step, float grad_scale, void* stream);
/* K18 helper, EXTENSION (the reference is single-device): bf16 wire format of the data-parallel gradient all-reduce -- a bucket of
 * the flat f32 gradient buffer is rounded to bf16 (round-to-nearest-even) for the collective and widened back into the f32 master
 * buffer afterwards (trainer.GradSync(wire_dtype="bf16")).  n a multiple of 8, buffers 16-byte aligned. */
int maavss_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
int maavss_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream);

/* ---- K1-K6 DINO ViT-S/8 attention-frame extractor ------------------------------------------------
 * What VideoAttention._inference (video_attention.py:38-103) obtains from dino's
 * VisionTransformer.get_last_selfattention (external module, call site video_attention.py:52), batched
 * over frames.  rows = frames * ntok, ntok = (H/8)*(W/8) + 1.
 * dtype    : the 16-bit storage / MFMA operand format of the activations and weights, raw uint16 bits: 0 = bf16,
 *            2 = IEEE half (same MFMA rate, 3 more mantissa bits; the LayerNorm-ed / softmax-bounded activations of the
 *            extractor fit its range).  "bf16" below stands for whichever is selected; accumulation, residual stream,
 *            LayerNorm and softmax are always f32.
 * patchify : frames [F][3][H][W] f32 -> a [rows][192] bf16 (CLS rows zero).
 * gemm     : C = epilogue(A[M][K] bf16 . W[N][K]^T bf16); N % 128 == 0, K % 64 == 0.  epilogue 0: +bias,
 *            columns < qscale_cols times qscale -> bf16;  1: +bias, exact GELU -> bf16;  2: C(f32) += acc + bias
 *            (residual, in place);  3: C(f32) = acc + table[row % period][N] (cls/pos-embed/conv-bias table).
 * layernorm: x [rows][384] f32 -> bf16 (eps as given, 1e-6 for DINO).
 * attn     : qkv [rows][ld_qkv] bf16 (q | k | v, heads x 64 each, q pre-scaled) -> out [rows][ld_out] bf16.
 * cls_attn : last block: softmax of the CLS query over all tokens, CLS column dropped -> att [F][heads][ntok-1] f32.
 * attn_maps: video_attention.py:80-96 + av_dataset.py:328: head sum, x(1/frame max), nearest x8 upsample,
 *            x(1/clip max over groups of clip_frames frames; 0 = skip) -> out [F][1][H][W] f32 (zero outside the
 *            patch grid); ws = F * ((H/8)*(W/8) + 1) floats. */
int maavss_vit_patchify(const float* frames, void* a, int64_t n_frames, int H, int W, int dtype, void* stream);
int maavss_vit_gemm(const void* A, int lda, const void* W, const float* bias, const float* table, int period, void* C,
                    int ldc, int64_t M, int N, int K, int epilogue, int qscale_cols, float qscale, int dtype, void* stream);
int maavss_vit_layernorm(const float* x, const float* gamma, const float* beta, void* y, int64_t rows, int dim,
                         float eps, int dtype, void* stream);
/* panel GEMM for the K = 384 layers with the preceding LayerNorm fused (dino Block: norm1->attn.qkv, attn.proj,
 * norm2->mlp.fc1): C = epilogue(LN(X)[M][384] . W[N][384]^T) when X (f32) is given, or A (bf16, [M][lda]) . W^T
 * otherwise; exactly one of X / A is non-null.  epilogue 0 / 1 / 2 as in maavss_vit_gemm (bf16+q-scale, bf16+GELU,
 * f32 residual in place).  N % 128 == 0, N <= 2048, ldc % 8 == 0, qscale_cols % 8 == 0.  C must be ALLOCATED with
 * c_rows >= ceil(M/128)*128 rows: the register epilogue stores whole 128-row panels unguarded (rows >= M receive
 * don't-care values) so that its counted DMA waits stay exact. */
int maavss_vit_panel_gemm(const float* X, const void* A, int lda, const float* ln_gamma, const float* ln_beta,
                          float ln_eps, const void* W, const float* bias, void* C, int ldc, int64_t c_rows, int64_t M,
                          int N, int epilogue, int qscale_cols, float qscale, int dtype, void* stream);
/* weight-stationary GEMM for the same K = 384 layers (attn.qkv, attn.proj, mlp.fc1 of dino's Block; call site
 * video_attention.py:52): C = epilogue(A[M][384] (16-bit, dense rows) . W[N][384]^T), the weights held in registers, the
 * activation rows streamed through LDS in 64-row panels.  epilogue 0 / 1 / 2 as in maavss_vit_gemm; epilogue 4 (ABI 400, dtype 2
 * only) = epilogue 1 with the GELU polynomial evaluated in packed IEEE half instead of f32 (mlp.fc1: 7 % faster, twice the
 * rounding error of the stored value -- VideoAttention(gelu="half"), never the default).  N % 384 == 0 (one
 * workgroup per 384 columns; the N / 384 workgroups of a row range share one XCD's L2), ldc % 8 == 0, qscale_cols % 384 == 0.
 * A and C must be ALLOCATED with a_rows, c_rows >= ceil(M/64)*64 rows (whole panels are read and stored; rows >= M hold
 * don't-care values).  xn_out (epilogue 2 with N = 384 only, may be null): additionally LayerNorm(ln_gamma, ln_beta, ln_eps)
 * of every updated row of C -> 16-bit [c_rows][384], so that the consumer GEMM needs no LayerNorm pass.  ln_gamma = ln_beta = NULL
 * (ABI 400; here and in maavss_vit_ws_gemm_ln / _ln_mx): the LayerNorm WITHOUT its affine part, (x - mean) * rstd -- for callers that
 * have folded gamma / beta into the consumer's frozen weights (W' = W diag(gamma), b' = b + W beta).  maavss_amd.VideoAttention does NOT fold by
 * default: measured +0.3 % clips/s for another rounding realisation of the weights (end-to-end mask-MSE 4.0e-6 ... 8.4e-6 instead of 5.4e-6 ...
 * 6.5e-6 over the three gated cases: same mean, thinner worst-case margin to 1e-5; HISTORY.md H4). */
int maavss_vit_ws_gemm(const void* A, int lda, int64_t a_rows, const void* W, const float* bias, void* C, int ldc,
                       int64_t c_rows, int64_t M, int N, int epilogue, int qscale_cols, float qscale, void* xn_out,
                       const float* ln_gamma, const float* ln_beta, float ln_eps, int dtype, void* stream);
/* norm1 -> attn.qkv without a LayerNorm pass: X f32 [x_rows][384] is normalised on the way into LDS (epilogue 0: +bias,
 * q-scale -> 16-bit).  row_stats [M][3][2] = (mean, sum of squared deviations) of the three 128-column thirds of every row of X,
 * as left by maavss_vit_gemm_stats when it wrote X. */
int maavss_vit_ws_gemm_ln(const float* X, int64_t x_rows, const float* row_stats, const float* ln_gamma, const float* ln_beta,
                          float ln_eps, const void* W, const float* bias, void* C, int ldc, int64_t c_rows, int64_t M, int N,
                          int qscale_cols, float qscale, int dtype, void* stream);
/* The same layer with the LayerNorm applied AFTER the product (round 4): the rows of X are only rounded to the storage format on the way in,
 * W is the gamma-folded weight W diag(gamma) (storage format), col_sums[n] = sum over k of that ROUNDED W's row n (f32), bias = b + W beta, and the
 * epilogue computes  rstd_r (x W^T - mean_r col_sums[n]) + bias[n]  (then the q-scale).  Equal to maavss_vit_ws_gemm_ln in exact arithmetic;
 * in 16 bits another rounding realisation (raw rows rounded instead of normalised ones).  No per-element LayerNorm arithmetic in the loader. */
int maavss_vit_ws_gemm_ln_post(const float* X, int64_t x_rows, const float* row_stats, const float* col_sums, float ln_eps, const void* W,
                               const float* bias, void* C, int ldc, int64_t c_rows, int64_t M, int N, int qscale_cols, float qscale, int dtype,
                               void* stream);
/* maavss_vit_gemm with the f32 epilogues (2, 3) additionally writing, for every row and every 128-column tile, (mean, sum of
 * squared deviations from that mean) of the values it stored: row_stats [M][N / 128][2] (null = maavss_vit_gemm). */
int maavss_vit_gemm_stats(const void* A, int lda, const void* W, const float* bias, const float* table, int period, void* C,
                          int ldc, int64_t M, int N, int K, int epilogue, int qscale_cols, float qscale, float* row_stats,
                          int dtype, void* stream);
int maavss_vit_attn(const void* qkv, void* out, int frames, int ntok, int heads, int ld_qkv, int ld_out, int dtype,
                    void* stream);
int maavss_vit_cls_attn(const void* qkv, float* att, int frames, int ntok, int heads, int ld_qkv, int dtype, void* stream);
/* Block-scaled fp8 attention, BASELINE config "fp8 MFMA attention QK^T / AV" (round 3; replaces round 2's non-scaled fp8 kernel) --
 * same call site (video_attention.py:52) and the out tensor / layout of maavss_vit_attn: Q K^T and P V on v_mfma_scale_f32_32x32x64_f8f6f4 (OCP MX:
 * e4m3 elements, one e8m0 scale per 32-element K block; f32 accumulation -- the config's "bf16 accumulate" is f32 here, wider).
 * `ws` (maavss_vit_attn_mx_ws_bytes(rows) bytes, 256-byte aligned, rows = frames * ntok) holds the operand images laid out in
 * csrc/vit_mx.h: q8 / k8 [rows_alloc][384] with per-(token, head, 32 d) scales, V transposed [384][rows_alloc] with per-(d row,
 * 32-token block) scales.  They are written either by maavss_vit_qkv_mx from a 16-bit qkv tensor, or directly by the attn.qkv
 * GEMM (maavss_vit_ws_gemm_ln_mx: no 16-bit qkv, no quantisation pass).  `dtype` = the 16-bit format of out (and of qkv). */
int64_t maavss_vit_attn_mx_ws_bytes(int64_t rows);
int maavss_vit_qkv_mx(const void* qkv, void* ws, int64_t rows, int ld_qkv, int dtype, void* stream);
int maavss_vit_attn_mx(const void* ws, void* out, int frames, int ntok, int heads, int ld_out, int dtype, void* stream);
/* attn.qkv (N = 1152) with norm1 applied on the way in, exactly as maavss_vit_ws_gemm_ln, but the epilogue writes the images above
 * into `mx_ws` (maavss_vit_attn_mx_ws_bytes(M) bytes; bytes past the stored panels must be zero -- allocate zeroed once) instead
 * of a 16-bit qkv tensor: no quantisation pass, half the output bytes.  `dtype` = the 16-bit format of W. */
int maavss_vit_ws_gemm_ln_mx(const float* X, int64_t x_rows, const float* row_stats, const float* ln_gamma, const float* ln_beta,
                             float ln_eps, const void* W, const float* bias, void* mx_ws, int64_t M, int qscale_cols, float qscale,
                             int dtype, void* stream);
int maavss_vit_attn_maps(const float* att, float* out, float* ws, int64_t n_frames, int heads, int H, int W,
                         int clip_frames, int attn_diff /* av_dataset.py:323-326, needs clip_frames > 0 */, void* stream);
/* Same, and sets *nonfinite_flag (device int32, sticky, never cleared here; null = no check) to 1 when any CLS-attention value
 * is inf or NaN.  This is the range guard of the IEEE-half storage format (dtype 2) of the extractor: the reference computes in
 * fp32 (video_attention.py:52), where |activation| > 65504 is harmless; here such a value becomes inf in a 16-bit store,
 * poisons that frame's residual stream and arrives in this row as NaN -- one check on 6 x n values per frame covers every
 * upstream store.  The host side (maavss_amd/video_attention.py) raises and names act_dtype="bf16". */
int maavss_vit_attn_maps_checked(const float* att, float* out, float* ws, int64_t n_frames, int heads, int H, int W,
                                 int clip_frames, int attn_diff, int32_t* nonfinite_flag, void* stream);

/* ---- EXTENSION (no reference counterpart): AdaptiveAvgPool2d closing the STFT encoder for frame sizes the
 * reference constructor cannot build (224^2, 384^2; SURVEY.md finding 2).  x NHWC [B][H][W][C]; out/dout
 * addressed b*os_b + (oy*Wo+ox)*os_p + c*os_c. */
int maavss_adaptive_pool_fwd(const float* x, float* out, int B, int H, int W, int C, int Ho, int Wo, int64_t os_b,
                             int64_t os_p, int64_t os_c, void* stream);
int maavss_adaptive_pool_bwd(const float* dout, float* dx, int B, int H, int W, int C, int Ho, int Wo, int64_t os_b,
                             int64_t os_p, int64_t os_c, void* stream);

#ifdef __cplusplus
}
#endif
#endif
