"""TEST INFRASTRUCTURE ONLY -- CPU oracle for SURVEY.md 8 row f1: the phasegram variant of the fusion network.

Restates, in plain torch fp32 on the CPU,
  * `avse_model.AV_Fusion_Model` (avse_model.py:410-711) -- what train_av_net.py:5,66-69 builds -- as `AVFusionRef`:
    same constructor arguments, same forward contract, same state_dict keys/shapes;
  * `utilities.video_phasegram` (utilities.py:206-228) as `video_phasegram_ref` (without the torchvision resize:
    torchvision is absent here, frames are taken at the phasegram size).
PINNED: oracle/make_golden.py imports the reference's own class in the build container, asserts that this twin
reproduces it (forward, loss, every gradient, BN statistics) and commits tests/golden/avfm_A.npz.
`video_phasegram_ref` is checked there against the reference's function too (it needs only torch/numpy).

The layer plans are computed analytically; the reference derives them by running dummy tensors through the layers as
it adds them (avse_model.py:428-444, 449-461, 480-503, 575-601).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

FUSED_DIM = 512        # avse_model.py:516
LSTM_HIDDEN = 256      # avse_model.py:545


def plan_pgram_encoder(h, w, latent, fc_size):
    """avse_model.py:428-444 -> [(c_in, c_out)] of Conv2d(k=(1,9), stride (1,2), padding (0,4)) + BN + Tanh; final width."""
    plan, c_in = [], 1
    while w * h * latent > fc_size // 2:
        if len(plan) > 32:
            raise ValueError("phasegram encoder does not converge")
        c_out = min(c_in * 2, latent)
        plan.append((c_in, c_out))
        w = (w + 8 - 9) // 2 + 1
        c_in = c_out
    return plan, w


def plan_pgram_decoder(w_enc, w_full, latent):
    """avse_model.py:449-461 -> [(c_in, c_out, followed_by_bn_tanh)] of ConvTranspose2d(k=(1,9), s (1,2), p (0,4), op (0,1))."""
    plan, c_in, w = [], latent, w_enc
    while w < w_full:
        c_out = max(c_in // 2, 1)
        w = (w - 1) * 2 - 8 + 9 + 1
        plan.append((c_in, c_out, w != w_full))
        c_in = c_out
    return plan


def plan_stft_encoder(t_a, n_bins, h, w_enc, latent):
    """avse_model.py:473-498 -> [(c_in, c_out, stride)] of Conv2d(k=5, padding 2) + BN + Tanh.  The reference tracks
    the map size by integer halving (exact for even sizes, which is all it is ever run with)."""
    plan, c_in, cur = [], 2, [t_a, n_bins]
    while cur != [h, w_enc]:
        if len(plan) > 32:
            raise ValueError("STFT encoder cannot reach the phasegram code's shape")
        c_out = min(c_in * 4, latent)
        stride = [1, 1]
        for d, tgt in ((0, h), (1, w_enc)):
            if cur[d] > tgt:
                stride[d] = 2
                cur[d] //= 2
        plan.append((c_in, c_out, tuple(stride)))
        c_in = c_out
    return plan


def plan_stft_decoder(t_a, n_bins, h, w_enc, latent, c_stft):
    """avse_model.py:575-601 -> [(c_in, c_out, stride, out_pad, followed_by_bn_tanh)] of ConvTranspose2d(k=5, p 2)."""
    plan, c_in, cur = [], latent, [h, w_enc]
    while cur != [t_a, n_bins]:
        if len(plan) > 32:
            raise ValueError("STFT decoder cannot reach the STFT shape")
        c_out = max(c_in // 4, c_stft)
        stride, opad = [1, 1], [0, 0]
        for d, full in ((0, t_a), (1, n_bins)):
            if cur[d] < full:
                stride[d], opad[d] = 2, 1
                cur[d] *= 2
        plan.append((c_in, c_out, tuple(stride), tuple(opad), cur != [t_a, n_bins]))
        c_in = c_out
    return plan


class AVFusionRef(nn.Module):
    def __init__(self, stft_shape, pgram_shape, alpha, latent_channels=64, fc_size=4096):
        super().__init__()
        self.stft_shape, self.pgram_shape, self.latent_channels = list(stft_shape), list(pgram_shape), latent_channels
        t_a, n_bins = stft_shape[-2], stft_shape[-1]
        h, w_full = pgram_shape[-2], pgram_shape[-1]
        enc_plan, w_enc = plan_pgram_encoder(h, w_full, latent_channels, fc_size)
        mods = []
        for ci, co in enc_plan:
            mods += [nn.Conv2d(ci, co, (1, 9), (1, 2), (0, 4)), nn.BatchNorm2d(co), nn.Tanh()]
        self.phasegram_encoder = nn.Sequential(*mods)
        mods = []
        for ci, co, bn in plan_pgram_decoder(w_enc, w_full, latent_channels):
            mods.append(nn.ConvTranspose2d(ci, co, (1, 9), (1, 2), (0, 4), (0, 1)))
            if bn:
                mods += [nn.BatchNorm2d(co), nn.Tanh()]
        self.phasegram_decoder = nn.Sequential(*mods)
        c_v = enc_plan[-1][1] if enc_plan else 1
        mods, c_a = [], 2
        for ci, co, st in plan_stft_encoder(t_a, n_bins, h, w_enc, latent_channels):
            mods += [nn.Conv2d(ci, co, (5, 5), st, (2, 2)), nn.BatchNorm2d(co), nn.Tanh()]
            c_a = co
        self.stft_encoder = nn.Sequential(*mods)
        self.lstm = nn.LSTM(input_size=(c_v + c_a) * w_enc, hidden_size=LSTM_HIDDEN, num_layers=1, bias=False,
                            batch_first=True, bidirectional=True)
        if h * 2 * LSTM_HIDDEN != fc_size:
            raise ValueError(f"fc1 expects {fc_size} inputs but the BiLSTM over {h} phasegram rows yields {h * 2 * LSTM_HIDDEN} "
                             f"(the reference fails in its own constructor, avse_model.py:550-553)")
        self.fc1 = nn.Linear(fc_size, fc_size // 2)
        self.fc2 = nn.Linear(fc_size // 2, FUSED_DIM)
        mods = []
        for ci, co, st, op, bn in plan_stft_decoder(t_a, n_bins, h, w_enc, latent_channels, stft_shape[1]):
            mods.append(nn.ConvTranspose2d(ci, co, (5, 5), st, (2, 2), op))
            if bn:
                mods += [nn.BatchNorm2d(co), nn.Tanh()]
        self.stft_decoder = nn.Sequential(*mods)
        self.stft_autoencoder = nn.Sequential(*self.stft_encoder, *self.stft_decoder)
        self.phasegram_autoencoder = nn.Sequential(*self.phasegram_encoder, *self.phasegram_decoder)
        self.a_fc1 = nn.Sequential(nn.Linear(FUSED_DIM, 2 * t_a * n_bins), nn.LeakyReLU(negative_slope=0.3))
        self.v_fc1 = nn.Sequential(nn.Linear(FUSED_DIM, h * w_full), nn.LeakyReLU(negative_slope=0.3))

    def av_fusion_forward(self, x_a, x_v):
        seq = torch.cat((x_v.permute(0, 2, 1, 3), x_a.permute(0, 2, 1, 3)), dim=2).flatten(-2, -1)
        av = self.lstm(seq)[0].flatten(1)
        av = F.leaky_relu(self.fc1(av), negative_slope=0.3)
        return F.leaky_relu(self.fc2(av), negative_slope=0.3)

    def visual_ae_forward(self, x_v):
        return self.phasegram_autoencoder(x_v)

    def audio_ae_forward(self, x_a):
        return self.stft_autoencoder(x_a)

    def forward(self, x_a, x_v):
        fused = self.av_fusion_forward(self.stft_encoder(x_a), self.phasegram_encoder(x_v))
        return self.a_fc1(fused).view(x_a.shape), self.v_fc1(fused).view(x_v.shape), fused


def canonical_key(module, key):
    """stft_autoencoder.* / phasegram_autoencoder.* alias the encoder / decoder modules."""
    for alias, enc, dec in (("stft_autoencoder.", "stft_encoder", "stft_decoder"),
                            ("phasegram_autoencoder.", "phasegram_encoder", "phasegram_decoder")):
        if key.startswith(alias):
            idx, rest = key[len(alias):].split(".", 1)
            n_enc = len(getattr(module, enc))
            return f"{enc}.{idx}.{rest}" if int(idx) < n_enc else f"{dec}.{int(idx) - n_enc}.{rest}"
    return key


def seeded_state_dict(module, seed):
    from oracle.avse_ref_cpu import seeded_tensor
    out = {}
    for k, v in module.state_dict().items():
        t = seeded_tensor(canonical_key(module, k), v.shape, seed)
        if k.rsplit(".", 1)[-1] == "bias" and v.dim() == 1 and "fc" in k:      # Linear biases: small, centred
            t = (t - t.mean()) * 0.1
        out[k] = t
    return out


def load_seeded(module, seed):
    module.load_state_dict(seeded_state_dict(module, seed), strict=True)
    return module


def video_phasegram_ref(frames, diff=True, cumulative=True, normalize=True):
    """utilities.py:206-228 without the resize: attention frames [B,1,T,H,W] -> phasegram [B,1,T,H*W].
    As in the reference the final normalisation divides by the maximum over the WHOLE batch tensor."""
    frames = torch.squeeze(frames, 1)
    fft = torch.fft.fftshift(torch.fft.fft2(frames))
    p_flat = torch.flatten(torch.angle(fft), start_dim=-2, end_dim=-1)
    if cumulative:
        p_flat = torch.cumsum(p_flat, dim=-1)
        p_flat = p_flat / (2.0 * math.pi * p_flat.shape[-1])
    else:
        p_flat = (p_flat + math.pi) / (math.pi * 2.0)
    if diff:
        p_diff = torch.diff(p_flat, dim=-2)
        pg = torch.cat((torch.zeros_like(p_diff[:, 0:1, :]), p_diff), dim=1)
    else:
        pg = p_flat
    pg = torch.unsqueeze(pg, 1)
    if normalize:
        pg = pg * (1.0 / torch.max(torch.abs(pg)))
    return pg
