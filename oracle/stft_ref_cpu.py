"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the audio STFT (+noise) path.

Follows the reference's
  utilities.py:24-28      calc_hop_size
  av_dataset.py:106       window = torch.hamming_window(fft_len)        (periodic)
  av_dataset.py:157-174   torchaudio.functional.spectrogram(pad=0, window, n_fft, hop, win_length=n_fft,
                          power=None, normalized, onesided=True), then drop the last frame
                          (and the last bin when trim_stft_end)
  av_dataset.py:335-342   permute to [2, T_a, F]; optional y *= 1/max(|y| + 1e-7); x = y + sigma * N(0,1)

PARITY UNPINNED: torchaudio (requirements.txt:3, no version pin) is not importable
in the container and its source is not under /root/reference.  Its published
`spectrogram` is: torch.stft(center=True, pad_mode="reflect", normalized=False,
onesided, return_complex) ; if normalized: /= window.pow(2).sum().sqrt().  That is
restated here on torch.stft, and `stft_direct_f64` is an independent float64
restatement (explicit reflect pad + framing + DFT matrix) that the tests check
the torch.stft form against.
"""
import math

import torch


def calc_hop_size(num_frames, hops_per_frame, fps, sr):
    hop = int((sr / fps) / hops_per_frame)
    n = int(hops_per_frame * hop * num_frames)
    return hop, n, n // hop


def hamming_periodic(n, dtype=torch.float32):
    k = torch.arange(n, dtype=torch.float64)
    return (0.54 - 0.46 * torch.cos(2 * math.pi * k / n)).to(dtype)


def stft_ref(audio, fft_len, hop, normalized=True, trim_stft_end=False):
    """audio [..., L] fp32 -> y_stft [..., 2, T_a, F] (re/im planes, frames, bins)."""
    win = torch.hamming_window(fft_len, dtype=audio.dtype)
    spec = torch.stft(audio, fft_len, hop_length=hop, win_length=fft_len, window=win, center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    if normalized:
        spec = spec / win.pow(2.0).sum().sqrt()
    spec = torch.view_as_real(spec)                 # [..., F, frames, 2]
    spec = spec[..., :-1, :-1, :] if trim_stft_end else spec[..., :, :-1, :]
    return spec.permute(*range(spec.dim() - 3), -1, -2, -3).contiguous()


def stft_direct_f64(audio, fft_len, hop, normalized=True):
    """Independent restatement: reflect pad, frame, window, DFT in float64."""
    a = audio.double()
    half = fft_len // 2
    left = a[..., 1:half + 1].flip(-1)
    right = a[..., -half - 1:-1].flip(-1)
    p = torch.cat([left, a, right], -1)
    n_frames = 1 + a.shape[-1] // hop
    idx = torch.arange(n_frames)[:, None] * hop + torch.arange(fft_len)[None, :]
    win = hamming_periodic(fft_len, torch.float64)
    fr = p[..., idx] * win                          # [..., frames, n]
    k = torch.arange(fft_len // 2 + 1, dtype=torch.float64)[:, None]
    n = torch.arange(fft_len, dtype=torch.float64)[None, :]
    ang = -2 * math.pi * k * n / fft_len
    re = fr @ torch.cos(ang).T
    im = fr @ torch.sin(ang).T
    out = torch.stack([re, im], -3)                 # [..., 2, frames, F]
    if normalized:
        out = out / win.pow(2).sum().sqrt()
    return out[..., :-1, :]


def gen_stft_example_ref(audio, fft_len, hop, sigma, noise, normalized=True, normalize_output=False):
    """av_dataset.py:335-342 with the Gaussian draw passed in (`noise` ~ N(0,1), same shape as y)."""
    y = stft_ref(audio, fft_len, hop, normalized)
    if normalize_output:
        amax = (y.abs() + 1e-7).flatten(-3).max(-1).values
        y = y * (1.0 / amax)[..., None, None, None]
    return y + noise * sigma, y


def synthetic_audio(batch, length, seed, sr=16000):
    """0.1*N(0,1) + sinusoids at 220/440/880 Hz (amp 0.2), clipped to [-1,1]."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    t = torch.arange(length, dtype=torch.float32) / sr
    a = 0.1 * torch.randn(batch, length, generator=g)
    for f in (220.0, 440.0, 880.0):
        ph = torch.rand(batch, 1, generator=g) * 2 * math.pi
        a = a + 0.2 * torch.sin(2 * math.pi * f * t[None, :] + ph)
    return a.clamp(-1, 1)


def istft_ref(stft, fft_len, hop, normalized=True, trim_stft_end=False):
    """AV_Dataset.istft (av_dataset.py:181-201), batched: stft [B,2,T_a,F] -> audio [B, hop*(T_a-1)].
    The reference hands torch.istft the real view [F,T,2] (accepted by the torch of its day); the same function takes
    the complex tensor today -- same arithmetic."""
    if trim_stft_end:
        stft = torch.nn.functional.pad(stft, (0, 1))
    spec = torch.complex(stft[:, 0], stft[:, 1]).transpose(1, 2).contiguous()          # [B, F, T]
    return torch.istft(spec, n_fft=fft_len, hop_length=hop, win_length=fft_len, window=hamming_periodic(fft_len),
                       normalized=normalized, onesided=True)
