"""Host-side mirror of the reference's STFT helpers on top of the K17 HIP kernel.

Mirrors utilities.calc_hop_size (utilities.py:24-28), AV_Dataset.stft (av_dataset.py:157-174),
AV_Dataset.gen_stft_example / add_noise (av_dataset.py:217-220, 335-342), batched over clips.
"""
import math

import torch

from . import _lib


def calc_hop_size(num_frames, hops_per_frame, fps, sr):
    hop = int((sr / fps) / hops_per_frame)
    audio_sample_len = int(hops_per_frame * hop * num_frames)
    return hop, audio_sample_len, audio_sample_len // hop


class STFT:
    """Batched STFT front end: audio [B, L] (cuda) -> (x_stft, y_stft) [B, 2, T_a, F]."""

    def __init__(self, fft_len, hop, normalized=True, trim_stft_end=False, noise_std=0.1,
                 normalize_output_fft=False, device="cuda"):
        self.fft_len, self.hop = fft_len, hop
        self.trim_stft_end = trim_stft_end
        self.noise_std = noise_std
        self.normalize_output_fft = normalize_output_fft
        k = torch.arange(fft_len, dtype=torch.float64)
        w = 0.54 - 0.46 * torch.cos(2 * math.pi * k / fft_len)        # torch.hamming_window (periodic)
        self.normalized = normalized
        self.raw_window = w.float().to(device)                         # synthesis window of inverse()
        if normalized:
            w = w / w.pow(2).sum().sqrt()
        self.window = w.float().to(device)

    def n_bins(self):
        return self.fft_len // 2 + (0 if self.trim_stft_end else 1)

    def __call__(self, audio, noise=None, seed=0, want_x=True):
        _lib.require_cuda(audio, noise)
        assert audio.dim() == 2 and audio.dtype == torch.float32 and audio.stride(1) == 1
        b, length = audio.shape
        n_frames = length // self.hop                     # 1 + L//hop frames minus the dropped last one
        f = self.n_bins()
        y = torch.empty(b, 2, n_frames, f, device=audio.device, dtype=torch.float32)
        x = torch.empty_like(y) if want_x else None
        amax = torch.zeros(b, device=audio.device, dtype=torch.float32) if self.normalize_output_fft else None
        if noise is not None:
            assert noise.shape == y.shape and noise.is_contiguous() and noise.dtype == torch.float32
        st = _lib.stream_ptr()
        direct_x = x if not self.normalize_output_fft else None
        _lib.call("maavss_stft_fwd", _lib.ptr(audio), b, length, audio.stride(0), _lib.ptr(self.window),
                  self.fft_len, self.hop, n_frames, f, _lib.ptr(y), _lib.ptr(direct_x), _lib.ptr(noise),
                  float(self.noise_std), int(seed), _lib.ptr(amax), st)
        if self.normalize_output_fft:
            _lib.call("maavss_stft_normalise", _lib.ptr(y), _lib.ptr(x), _lib.ptr(noise), _lib.ptr(amax), b,
                      n_frames, f, float(self.noise_std), int(seed), st)
        return x, y

    def inverse(self, stft):
        """AV_Dataset.istft (av_dataset.py:181-201): [2, T_a, F] or [B, 2, T_a, F] (cuda) -> audio [hop*(T_a-1)] /
        [B, hop*(T_a-1)].  As in the reference, the Nyquist bin trimmed by `trim_stft_end` is padded back with zeros and
        the scaling is torch.istft's `normalized` (frame_length ** 0.5), not the window-energy one the forward applies."""
        _lib.require_cuda(stft)
        single = stft.dim() == 3
        spec = (stft.unsqueeze(0) if single else stft).contiguous().float()
        b, two, n_frames, f = spec.shape
        assert two == 2 and f == self.n_bins(), f"expected [.., 2, T, {self.n_bins()}], got {tuple(stft.shape)}"
        out_len = self.hop * (n_frames - 1)
        frames = torch.empty(b, n_frames, self.fft_len, device=spec.device, dtype=torch.float32)
        audio = torch.empty(b, out_len, device=spec.device, dtype=torch.float32)
        _lib.call("maavss_istft", _lib.ptr(spec), b, n_frames, f, _lib.ptr(self.raw_window), self.fft_len, self.hop,
                  1 if self.normalized else 0, _lib.ptr(frames), _lib.ptr(audio), audio.stride(0), _lib.stream_ptr())
        return audio[0] if single else audio
