"""TEST INFRASTRUCTURE ONLY -- fp32 CPU oracle for the AV fusion network.

Restates, in plain torch.nn, what the reference computes in
  avse_model_final.py:33-59   (visual encoder: 5x Conv3d(3,5,5)+BN3d+MaxPool+LeakyReLU)
  avse_model_final.py:75-107  (STFT encoder: Conv2d(3,9)+BN2d+Tanh until [T_a,F]->[T,S])
  avse_model_final.py:124-146 (BiLSTM over the 16 channel steps, fc1, fc2)
  avse_model_final.py:155-200 (ConvTranspose2d decoder mirror + autoencoder alias)
  avse_model_final.py:203-213 (output heads)
  avse_model_final.py:235-274 (forward)
  train_avse_frames.py:164-181 (loss, backward, Adam)

Unlike the reference constructor this one derives every shape analytically (no
dry-run tensors, no "cuda" device string, no prints, no RNG side effects) and
raises where the reference's `while` loop would never terminate (224^2 / 384^2,
SURVEY.md finding 2).  Module names are kept so that state_dict keys are the
reference's keys.  PINNED by oracle/make_golden.py against the
imported reference; see tests/golden/.

`spatial_match="adaptive"` is an EXTENSION that is not in the reference: it ends
the STFT encoder with an adaptive average pool so that frame sizes such as 224^2
(S=9) become constructible; it has no reference counterpart, hence no pin.
"""
import zlib

import torch
import torch.nn as nn
import torch.nn.functional as F

LSTM_HIDDEN = 256
FUSED_DIM = 512


def visual_side(width):
    """Side of the flattened spatial map after the visual encoder
    (avse_model_final.py:33-58): three /2 pools, one /3 pool, a pad-3 conv that
    grows the map by 2, one more /3 pool."""
    s = width
    for _ in range(3):
        s = s // 2
    s = s // 3
    s = s + 2
    return s // 3


def plan_stft_encoder(t_a, n_bins, t_v, s_v, latent, spatial_match="exact"):
    """Layer plan of the STFT encoder (avse_model_final.py:82-105).

    Returns (layers, pool) with layers = [(c_in, c_out, (sh, sw), (ph, pw))] and
    pool = None or the adaptive-pool target (extension, not in the reference)."""
    cur = [t_a, n_bins]
    tgt = [t_v, s_v]
    layers = []
    c_in = 2

    def halvable(d):
        if spatial_match == "exact":
            return cur[d] > tgt[d]
        return cur[d] > tgt[d] and cur[d] // 2 >= tgt[d]

    while True:
        if spatial_match == "exact":
            if cur == tgt:
                break
            if cur[0] < tgt[0] or cur[1] < tgt[1] or len(layers) > 16:
                raise ValueError(
                    f"STFT encoder cannot reach {tgt} from [{t_a}, {n_bins}] by halving "
                    f"(the reference constructor loops forever here); use a frame size in "
                    f"[96,167], [240,311] or [528,599], or spatial_match='adaptive'")
        elif not (halvable(0) or halvable(1) or c_in < latent):
            break
        c_out = min(c_in * 2, latent)
        stride = [1, 1]
        for d in (0, 1):
            if halvable(d):
                stride[d] = 2
                cur[d] = cur[d] // 2
        pad = (1, 3) if not layers else (1, 4)
        layers.append((c_in, c_out, tuple(stride), pad))
        c_in = c_out
    pool = None if cur == tgt else tuple(tgt)
    return layers, pool


def conv2d_out(hw, stride, pad, k=(3, 9)):
    return [(hw[d] + 2 * pad[d] - k[d]) // stride[d] + 1 for d in (0, 1)]


def plan_stft_decoder(t_a, n_bins, t_v, s_v, latent, c_stft=2):
    """ConvTranspose2d mirror (avse_model_final.py:155-193).  Returns
    [(c_in, c_out, (kh, kw), (sh, sw), (oph, opw), has_bn_tanh)]."""
    tracked = [t_v, s_v]
    actual = [t_v, s_v]
    k = [3, 9]
    c_in = latent
    plan = []
    while actual != [t_a, n_bins]:
        if len(plan) > 16:
            raise ValueError("STFT decoder cannot reach the STFT shape")
        c_out = max(c_in // 2, c_stft)
        stride = [1, 1]
        opad = [0, 0]
        for d, full in ((0, t_a), (1, n_bins)):
            if tracked[d] < full:
                stride[d] = 2
                opad[d] = 1
                tracked[d] *= 2
        pad = (1, 4)
        actual = [(actual[d] - 1) * stride[d] - 2 * pad[d] + k[d] + opad[d] for d in (0, 1)]
        this_k = tuple(k)
        k = [3, 9]
        if actual[1] == (n_bins - 1) // 2:
            k[1] = 10
        last = actual == [t_a, n_bins]
        plan.append((c_in, c_out, this_k, tuple(stride), tuple(opad), not last))
        c_in = c_out
    return plan


class _Conv3d16bit(torch.autograd.Function):
    """Conv3d whose operands are rounded where the HIP 16-bit path rounds them (maavss_amd/avse.py, precise=False):
    forward x and w to IEEE half, backward dy / w (input gradient) and dy / x (weight gradient) to bf16; products and
    sums in f32.  What remains between this and the kernels is summation order."""

    @staticmethod
    def forward(ctx, x, w, pad):
        ctx.save_for_backward(x, w)
        ctx.pad = pad
        return F.conv3d(x.half().float(), w.half().float(), padding=(1, pad, pad))

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        bf = lambda t: t.to(torch.bfloat16).float()     # noqa: E731
        dyb, pad = bf(dy), (1, ctx.pad, ctx.pad)
        dx = torch.nn.grad.conv3d_input(x.shape, bf(w), dyb, padding=pad) if ctx.needs_input_grad[0] else None
        dw = torch.nn.grad.conv3d_weight(bf(x), w.shape, dyb, padding=pad) if ctx.needs_input_grad[1] else None
        return dx, dw, None


class _Conv3dEmu(nn.Conv3d):
    def forward(self, x):
        return _Conv3d16bit.apply(x, self.weight, self.padding[1])


class _Conv3d16bitL0(torch.autograd.Function):
    """layer 0 (C_in = 1) of the HIP 16-bit path: IEEE-half operands on the MFMA in the forward pass; weight gradient with x and
    dy rounded to bf16 (conv3d_c1_wgrad_mfma_kernel); the network input needs no gradient"""

    @staticmethod
    def forward(ctx, x, w, pad):
        ctx.save_for_backward(x, w)
        ctx.pad = pad
        return F.conv3d(x.half().float(), w.half().float(), padding=(1, pad, pad))

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        bf = lambda t: t.to(torch.bfloat16).float()     # noqa: E731
        pad = (1, ctx.pad, ctx.pad)
        dx = torch.nn.grad.conv3d_input(x.shape, w, dy, padding=pad) if ctx.needs_input_grad[0] else None
        dw = torch.nn.grad.conv3d_weight(bf(x), w.shape, bf(dy), padding=pad) if ctx.needs_input_grad[1] else None
        return dx, dw, None


class _Conv3dEmuFwd(nn.Conv3d):
    def forward(self, x):
        return _Conv3d16bitL0.apply(x, self.weight, self.padding[1])


class AVFusionFramesRef(nn.Module):
    """Oracle twin of the reference's AV_Fusion_Model_Frames (same ctor args,
    same forward contract, same state_dict keys).  `emulate_16bit=True` (not in the reference) rounds the operands
    of the four C_in > 1 Conv3d layers like the HIP path's default 16-bit mode does -- used to separate quantisation
    error (vs this class with emulate_16bit=False, the pinned fp32 oracle) from implementation error."""

    def __init__(self, stft_shape, frame_shape, hops_per_frame, latent_channels=16, fc_size=4096,
                 spatial_match="exact", emulate_16bit=False):
        super().__init__()
        self.stft_shape = list(stft_shape)
        self.frame_shape = list(frame_shape)
        self.frame_channels = frame_shape[1]
        self.latent_channels = latent_channels
        self.output_stft_frames = hops_per_frame
        t_v, width = frame_shape[2], frame_shape[-1]
        t_a, n_bins = stft_shape[-2], stft_shape[-1]
        side = visual_side(width)
        s_v = side * side
        if side < 1:
            raise ValueError("frame too small for the visual encoder")

        chans = [1, 16, 32, 64, 64, latent_channels]
        pools = [2, 2, 2, 3, 3]
        pads = [2, 2, 2, 2, 3]
        mods = []
        for i in range(5):
            conv = (_Conv3dEmu if i > 0 else _Conv3dEmuFwd) if emulate_16bit else nn.Conv3d
            mods += [conv(chans[i], chans[i + 1], (3, 5, 5), 1, (1, pads[i], pads[i]), bias=False),
                     nn.BatchNorm3d(chans[i + 1]),
                     nn.MaxPool3d((1, pools[i], pools[i])),
                     nn.LeakyReLU()]
        mods.append(nn.Flatten(-2, -1))
        self.visual_encoder = nn.Sequential(*mods)

        layers, pool = plan_stft_encoder(t_a, n_bins, t_v, s_v, latent_channels, spatial_match)
        mods = []
        for (ci, co, st, pd) in layers:
            mods += [nn.Conv2d(ci, co, (3, 9), st, pd, bias=False), nn.BatchNorm2d(co), nn.Tanh()]
        if pool is not None:
            mods.append(nn.AdaptiveAvgPool2d(pool))
        self.stft_encoder = nn.Sequential(*mods)
        enc_ch = layers[-1][1]
        if enc_ch != latent_channels:
            raise ValueError(f"latent_channels={latent_channels} but the STFT encoder ends with {enc_ch} "
                             f"channels; torch.cat at avse_model_final.py:124 fails in the reference too")

        self.lstm = nn.LSTM(input_size=2 * t_v * s_v, hidden_size=LSTM_HIDDEN, num_layers=1, bias=False,
                            batch_first=True, bidirectional=True)
        flat = latent_channels * 2 * LSTM_HIDDEN     # the ctor argument fc_size is overwritten (:140)
        self.fc1 = nn.Linear(flat, flat // 2, bias=False)
        self.fc2 = nn.Linear(flat // 2, FUSED_DIM, bias=False)

        mods = []
        if pool is None:
            for (ci, co, k, st, op, bn) in plan_stft_decoder(t_a, n_bins, t_v, s_v, latent_channels, stft_shape[1]):
                mods.append(nn.ConvTranspose2d(ci, co, k, st, (1, 4), op, bias=False))
                if bn:
                    mods += [nn.BatchNorm2d(co), nn.Tanh()]
        self.stft_decoder = nn.Sequential(*mods)
        self.stft_autoencoder = nn.Sequential(*self.stft_encoder, *self.stft_decoder)

        self.a_fc1 = nn.Sequential(nn.Linear(FUSED_DIM, 2 * hops_per_frame * n_bins, bias=False))
        self.v_fc1 = nn.Sequential(nn.Linear(FUSED_DIM, self.frame_channels * width * width, bias=False))

    def av_fusion_forward(self, x_a, x_v):
        seq = torch.cat((x_v, x_a), dim=2).flatten(-2, -1)
        out, _ = self.lstm(seq)
        h = torch.tanh(self.fc1(out.flatten(1)))
        return torch.tanh(self.fc2(h))

    def audio_ae_forward(self, x_a):
        return self.stft_autoencoder(x_a)

    def forward(self, x_a, x_v):
        fused = self.av_fusion_forward(self.stft_encoder(x_a), self.visual_encoder(x_v))
        a = torch.tanh(self.a_fc1(fused)).view(x_a.shape[0], 2, self.output_stft_frames, x_a.shape[-1])
        v = torch.sigmoid(self.v_fc1(fused)).view(x_v.shape[0], self.frame_channels,
                                                  self.frame_shape[-2], self.frame_shape[-1])
        return a, v, fused


# ----------------------------------------------------------------------------------------------
# seeded weight recipe: every tensor of the state_dict is generated from a torch CPU generator
# seeded by crc32(key) ^ seed, so the 73 M weights are never committed, only regenerated.
# ----------------------------------------------------------------------------------------------
def seeded_tensor(key, shape, seed, dtype=torch.float32):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    u = torch.rand(tuple(shape), generator=g, dtype=torch.float32)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(tuple(shape), dtype=torch.long)
    if leaf == "running_var":
        return (0.5 + u).to(dtype)
    if leaf == "running_mean":
        return (0.2 * u - 0.1).to(dtype)
    if len(shape) == 1:                       # BN affine
        if leaf == "weight":
            return (0.5 + u).to(dtype)
        return (0.4 * u - 0.2).to(dtype)
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    bound = (3.0 / fan_in) ** 0.5 * 1.5
    return ((2 * u - 1) * bound).to(dtype)


def seeded_state_dict(module, seed):
    sd = module.state_dict()
    out = {}
    for k, v in sd.items():
        # stft_autoencoder.* aliases stft_encoder.* / stft_decoder.*; generate from the canonical key
        out[k] = seeded_tensor(canonical_key(module, k), v.shape, seed)
    return out


def canonical_key(module, key):
    if not key.startswith("stft_autoencoder."):
        return key
    idx, rest = key[len("stft_autoencoder."):].split(".", 1)
    idx = int(idx)
    n_enc = len(module.stft_encoder)
    if idx < n_enc:
        return f"stft_encoder.{idx}.{rest}"
    return f"stft_decoder.{idx - n_enc}.{rest}"


def load_seeded(module, seed):
    module.load_state_dict(seeded_state_dict(module, seed), strict=True)
    return module


def synthetic_batch(batch, t_v, width, t_a, n_bins, hops_per_frame, seed):
    """Seeded inputs/targets shaped like AV_Dataset.__getitem__ collated
    (av_dataset.py:333) and the window targets of train_avse_frames.py:152-162."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    x_a = torch.randn(batch, 2, t_a, n_bins, generator=g) * 0.5
    x_v = torch.rand(batch, 1, t_v, width, width, generator=g)
    y_a = torch.randn(batch, 2, hops_per_frame, n_bins, generator=g) * 0.3
    y_v = torch.rand(batch, 1, width, width, generator=g)
    return x_a, x_v, y_a, y_v


def loss_ref(model, x_a, x_v, y_a, y_v, loss_coeff=0.001, num_seq=1):
    """train_avse_frames.py:164-170."""
    a, v, fused = model(x_a, x_v)
    a_loss = F.mse_loss(a, y_a)
    v_loss = F.mse_loss(v, y_v)
    loss = (a_loss + loss_coeff * v_loss) / num_seq
    return loss, a_loss, v_loss, (a, v, fused)
