"""Drop-in `AV_Fusion_Model_Frames` (reference avse_model_final.py:14) on the MI355X HIP kernels.

Same constructor signature, same `forward(x_a, x_v) -> (x_a_out, x_v_out, x_av_fused)`, same
state_dict keys and tensor shapes as the reference, so a trainer only swaps the import
(INTEGRATION.md).  The torch.nn modules created here are PARAMETER HOLDERS (they give the reference's
key names, default initialisation and `.to()/.parameters()/.state_dict()` behaviour); their own
forward() is never called.  All arithmetic -- forward and the hand-written backward -- goes through
libmaavss_hip.so (maavss_amd.ops); there is no torch fallback and CPU tensors raise.

Differences from the reference constructor, all documented in DESIGN.md:
  * shapes are derived analytically (no dry-run tensors, no hard-coded "cuda", no prints, no RNG use);
  * frame sizes for which the reference's `while` loop never terminates raise ValueError, unless
    `spatial_match="adaptive"` (keyword-only EXTENSION: closes the STFT encoder with an adaptive average
    pool; needed for the 224^2 / 384^2 configurations of BASELINE.json);
  * `precise` (keyword-only): False (default, the bench path) = 16-bit MFMA operands with f32 accumulation in
    the conv3d layers -- IEEE half in the forward pass (activations are BatchNorm-bounded; 3 more mantissa
    bits than bf16 at the same MFMA rate keep the mask-MSE at ~2e-7), bf16 in the backward pass (gradients
    need the exponent range); True = exact-f32 MFMA everywhere (parity path).  The Linear/LSTM layers are
    weight-streaming (HBM-bound) and always use exact-f32 MFMA.
"""
import os

import torch
import torch.nn as nn

from . import _lib, ops

LSTM_HIDDEN = 256
FUSED_DIM = 512
_VIS_CH = (1, 16, 32, 64, 64)
_VIS_POOL = (2, 2, 2, 3, 3)
_VIS_PAD = (2, 2, 2, 2, 3)


def visual_spatial_side(width):
    """avse_model_final.py:33-58: /2 /2 /2 /3, the pad-3 conv adds 2, /3."""
    side = width // 2 // 2 // 2 // 3
    return (side + 2) // 3


def stft_encoder_plan(t_a, n_bins, t_v, s_v, latent, spatial_match):
    """avse_model_final.py:82-105 -> ([(c_in, c_out, stride, pad_w)], adaptive_pool_target | None)."""
    have, want = [t_a, n_bins], [t_v, s_v]
    plan, c_in = [], 2
    exact = spatial_match == "exact"
    while True:
        can = [have[d] > want[d] and (exact or have[d] // 2 >= want[d]) for d in (0, 1)]
        if exact:
            if have == want:
                break
            if have[0] < want[0] or have[1] < want[1] or len(plan) > 16:
                raise ValueError(
                    f"cannot halve the STFT [{t_a}, {n_bins}] down to the visual code [{t_v}, {s_v}]: the reference "
                    f"constructor (avse_model_final.py:82) never terminates for this frame size; buildable sizes are "
                    f"96-167, 240-311, 528-599 px, or pass spatial_match='adaptive'")
        elif not (can[0] or can[1] or c_in < latent):
            break
        c_out = min(2 * c_in, latent)
        stride = tuple(2 if can[d] else 1 for d in (0, 1))
        have = [have[d] // 2 if can[d] else have[d] for d in (0, 1)]
        plan.append((c_in, c_out, stride, 3 if not plan else 4))
        c_in = c_out
    return plan, (None if have == want else tuple(want))


def stft_decoder_plan(t_a, n_bins, t_v, s_v, latent, c_stft):
    """avse_model_final.py:155-193 -> [(c_in, c_out, kernel, stride, out_pad, followed_by_bn_tanh)]."""
    tracked, size = [t_v, s_v], [t_v, s_v]
    kernel, c_in, plan = (3, 9), latent, []
    while size != [t_a, n_bins]:
        if len(plan) > 16:
            raise ValueError("STFT decoder cannot reach the STFT shape")
        c_out = max(c_in // 2, c_stft)
        grow = [tracked[0] < t_a, tracked[1] < n_bins]
        stride = tuple(2 if g else 1 for g in grow)
        opad = tuple(1 if g else 0 for g in grow)
        tracked = [tracked[d] * (2 if grow[d] else 1) for d in (0, 1)]
        size = [(size[d] - 1) * stride[d] - 2 * (1, 4)[d] + kernel[d] + opad[d] for d in (0, 1)]
        plan.append((c_in, c_out, kernel, stride, opad, size != [t_a, n_bins]))
        kernel = (3, 10) if size[1] == (n_bins - 1) // 2 else (3, 9)
        c_in = c_out
    return plan


def _bn_eval_reduce(sums):
    """BatchNorm backward for a forward that used RUNNING statistics: mean and variance do not depend on the batch, so
    dy = gamma * invstd * g without the batch-mean terms.  The split backward (ops.bn_pool_act_bwd with `reduce_fn`) takes the
    dx coefficients from this [2C + 1] vector (sum g, sum g * xhat, count) and dgamma / dbeta from the untouched local copy:
    zeroing the two sums is the eval-mode formula."""
    sums[:-1].zero_()
    return sums


def _mark_touched(model, grads):
    """tell a FusedAdam built on this model which parameters just received a gradient (torch.optim.Adam skips the rest)"""
    flat = getattr(model, "_maavss_flat", None)
    if flat is not None:
        flat.mark(n for n, g in grads.items() if g is not None)


class _AVSEFunction(torch.autograd.Function):
    """One autograd node for the whole network: forward and backward are the HIP engine below."""

    @staticmethod
    def forward(ctx, model, x_a, x_v, *params):
        outs, saved = model._engine_forward(x_a, x_v, train=model.training)
        ctx.model, ctx.saved, ctx.was_training = model, saved, model.training
        return outs

    @staticmethod
    def backward(ctx, d_a, d_v, d_fused):
        model = ctx.model
        names = model._param_names
        need = {n: ctx.needs_input_grad[3 + i] for i, n in enumerate(names)}
        # eval-mode forward: BatchNorm normalised with its running statistics, which do not depend on the batch -- its
        # backward has no batch-mean terms (torch autograd allows this; fine-tuning with frozen statistics)
        grads = model._engine_backward(ctx.saved, d_a, d_v, d_fused, need, bn_eval=not ctx.was_training)
        ctx.saved = None
        _mark_touched(model, grads)
        return (None, None, None) + tuple(grads.get(n) for n in names)


_FUSION_PARAMS = ("lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.weight_ih_l0_reverse", "lstm.weight_hh_l0_reverse",
                  "fc1.weight", "fc2.weight")


class _FusionFunction(torch.autograd.Function):
    """av_fusion_forward (avse_model_final.py:235-251) from given encodings, as one autograd node."""

    @staticmethod
    def forward(ctx, model, x_a, x_v, *params):
        b, l, ts = x_a.shape[0], model.latent_channels, model.t_v * model.s_v
        # torch.cat((x_v, x_a), dim=2) + flatten == two strided device copies into the LSTM sequence buffer
        seq = torch.empty(b, l, 2 * ts, device=x_a.device, dtype=torch.float32)
        seq[:, :, :ts].copy_(x_v.reshape(b, l, ts))
        seq[:, :, ts:].copy_(x_a.reshape(b, l, ts))
        sv = {}
        fused = model._fusion_fwd(seq, sv)
        ctx.model, ctx.saved, ctx.shape = model, sv, tuple(x_a.shape)
        return fused

    @staticmethod
    def backward(ctx, d_fused):
        model, sv = ctx.model, ctx.saved
        grads = {}

        def wgrad_gemm(name, dz, x):
            if ctx.needs_input_grad[3 + _FUSION_PARAMS.index(name)]:
                grads[name] = ops.gemm(dz, x, trans_a=True, trans_b=True, precise=ops.MODE_F32)

        dgx = model._fusion_bwd(sv, d_fused.contiguous().float(), wgrad_gemm)
        d_a = d_v = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            b, ts = ctx.shape[0], model.t_v * model.s_v
            dseq = model._fusion_dseq(dgx, b)
            d_v = dseq[:, :, :ts].reshape(ctx.shape) if ctx.needs_input_grad[2] else None
            d_a = dseq[:, :, ts:].reshape(ctx.shape) if ctx.needs_input_grad[1] else None
        ctx.saved = None
        _mark_touched(model, grads)
        return (None, d_a, d_v) + tuple(grads.get(n) for n in _FUSION_PARAMS)


class _AEFunction(torch.autograd.Function):
    """audio_ae_forward (avse_model_final.py:254-256): stft_encoder -> stft_decoder as one autograd node."""

    @staticmethod
    def forward(ctx, model, x_a, *params):
        out, saved = model._ae_forward(x_a, train=model.training)
        ctx.model, ctx.saved, ctx.was_training = model, saved, model.training
        return out

    @staticmethod
    def backward(ctx, d_out):
        model = ctx.model
        grads = model._ae_backward(ctx.saved, d_out.contiguous().float(), bn_eval=not ctx.was_training)
        ctx.saved = None
        _mark_touched(model, {n: g for i, (n, g) in enumerate((n, grads.get(n)) for n in model._ae_param_names)
                              if ctx.needs_input_grad[2 + i]})
        return (None, None) + tuple(grads.get(n) if ctx.needs_input_grad[2 + i] else None
                                    for i, n in enumerate(model._ae_param_names))


class AV_Fusion_Model_Frames(nn.Module):
    def __init__(self, stft_shape, frame_shape, hops_per_frame, latent_channels=16, fc_size=4096, *,
                 spatial_match="exact", precise=False):
        super().__init__()
        self.stft_shape = list(stft_shape)
        self.frame_shape = list(frame_shape)
        self.frame_channels = frame_shape[1]
        self.latent_channels = latent_channels
        self.output_stft_frames = hops_per_frame
        self.precise = bool(precise)
        # The two halves of `precise`, separately switchable for parity isolation (tests/test_parity_r4_gpu.py: exact-f32 forward with
        # the bf16 backward gates the 16-bit backward kernels at model level; the 16-bit forward's MaxPool / LeakyReLU re-routing is
        # then out of the picture).  `precise=` sets both; product code never sets them apart.
        self.precise_fwd = self.precise_bwd = self.precise
        # 16-bit path: the first layer's conv output is recomputed instead of stored (MAAVSS_C1_RECOMPUTE=0: the storing kernels, for A/B)
        self.c1_recompute = os.environ.get("MAAVSS_C1_RECOMPUTE", "1") != "0"
        self._bn_sync = None
        if self.frame_channels != 1:
            raise ValueError("the visual encoder takes single-channel attention frames (avse_model_final.py:34)")
        if latent_channels not in (16,):
            # the reference only works when latent_channels equals the STFT encoder's last width (16)
            raise ValueError("latent_channels must be 16: torch.cat at avse_model_final.py:124 fails otherwise "
                             "in the reference as well (run_config.py's default 64 crashes there)")
        self.t_v, self.width = frame_shape[2], frame_shape[-1]
        if frame_shape[-2] != self.width:
            raise ValueError("square frames expected")
        self.t_a, self.n_bins = stft_shape[-2], stft_shape[-1]
        self.side = visual_spatial_side(self.width)
        if self.side < 1:
            raise ValueError("frame too small for the visual encoder")
        self.s_v = self.side * self.side

        chans = _VIS_CH + (latent_channels,)
        mods = []
        for i in range(5):
            mods += [nn.Conv3d(chans[i], chans[i + 1], (3, 5, 5), 1, (1, _VIS_PAD[i], _VIS_PAD[i]), bias=False),
                     nn.BatchNorm3d(chans[i + 1]), nn.MaxPool3d((1, _VIS_POOL[i], _VIS_POOL[i])), nn.LeakyReLU()]
        mods.append(nn.Flatten(-2, -1))
        self.visual_encoder = nn.Sequential(*mods)

        self._enc_plan, self._enc_pool = stft_encoder_plan(self.t_a, self.n_bins, self.t_v, self.s_v,
                                                           latent_channels, spatial_match)
        mods = []
        for (ci, co, st, pw) in self._enc_plan:
            mods += [nn.Conv2d(ci, co, (3, 9), st, (1, pw), bias=False), nn.BatchNorm2d(co), nn.Tanh()]
        if self._enc_pool is not None:
            mods.append(nn.AdaptiveAvgPool2d(self._enc_pool))
        self.stft_encoder = nn.Sequential(*mods)
        if self._enc_plan[-1][1] != latent_channels:
            raise ValueError("STFT encoder does not end with latent_channels channels")

        self.seq_in = 2 * self.t_v * self.s_v
        self.lstm = nn.LSTM(input_size=self.seq_in, hidden_size=LSTM_HIDDEN, num_layers=1, bias=False,
                            batch_first=True, bidirectional=True)
        flat = latent_channels * 2 * LSTM_HIDDEN          # ctor arg fc_size is overwritten (avse_model_final.py:140)
        self.fc1 = nn.Linear(flat, flat // 2, bias=False)
        self.fc2 = nn.Linear(flat // 2, FUSED_DIM, bias=False)

        mods = []
        self._dec_plan = []
        if self._enc_pool is None:
            self._dec_plan = stft_decoder_plan(self.t_a, self.n_bins, self.t_v, self.s_v, latent_channels, stft_shape[1])
            for (ci, co, k, st, op, bn) in self._dec_plan:
                mods.append(nn.ConvTranspose2d(ci, co, k, st, (1, 4), op, bias=False))
                if bn:
                    mods += [nn.BatchNorm2d(co), nn.Tanh()]
        self.stft_decoder = nn.Sequential(*mods)
        self.stft_autoencoder = nn.Sequential(*self.stft_encoder, *self.stft_decoder)

        self.a_fc1 = nn.Sequential(nn.Linear(FUSED_DIM, 2 * hops_per_frame * self.n_bins, bias=False))
        self.v_fc1 = nn.Sequential(nn.Linear(FUSED_DIM, self.frame_channels * self.width * self.width, bias=False))

        # parameters that take part in forward(), in a fixed order (stft_decoder.* get no gradient, like the reference)
        self._param_names = [n for n, _ in self.named_parameters()
                             if not n.startswith("stft_decoder.") and not n.startswith("stft_autoencoder.")]
        # parameters of audio_ae_forward (named_parameters lists shared modules once, under their first name)
        self._ae_param_names = [n for n, _ in self.named_parameters()
                                if n.startswith("stft_encoder.") or n.startswith("stft_decoder.")]

    # ---- reference API: gradient toggles (avse_model_final.py:216-232) --------------------------------
    def toggle_fusion_grads(self, toggle):
        for m in (self.lstm, self.fc1, self.fc2, self.a_fc1, self.v_fc1):
            m.requires_grad_(toggle)

    def toggle_stft_ae_grads(self, toggle):
        for m in self.stft_autoencoder:
            m.requires_grad_(toggle)

    def toggle_enc_grads(self, toggle):
        for m in self.stft_encoder:
            m.requires_grad_(toggle)
        for m in self.visual_encoder:
            m.requires_grad_(toggle)

    def audio_ae_forward(self, x_a):
        """stft_autoencoder(x_a): [B,2,T_a,F] -> [B,2,T_a,F] (avse_model_final.py:254-256; train_audio_net.py:108)."""
        if self._enc_pool is not None:
            raise NotImplementedError("the STFT autoencoder is only defined for the reference's exact shape matching "
                                      "(spatial_match='exact'): the 'adaptive' extension has no decoder")
        _lib.require_cuda(x_a)
        pd = dict(self.named_parameters())
        return _AEFunction.apply(self, x_a, *[pd[n] for n in self._ae_param_names])

    # ---- forward ----------------------------------------------------------------------------------------
    def forward(self, x_a, x_v):
        _lib.require_cuda(x_a, x_v)
        pd = dict(self.named_parameters())
        params = [pd[n] for n in self._param_names]
        return _AVSEFunction.apply(self, x_a, x_v, *params)

    def av_fusion_forward(self, x_a, x_v):
        """avse_model_final.py:235-251: encodings x_a, x_v [B,16,T,S] -> x_av_fused [B,512] (cat on dim 2, flatten,
        BiLSTM over the 16 channel steps, fc1, tanh, fc2, tanh) as one autograd node over the HIP engine;
        differentiable in both encodings and the LSTM / fc weights."""
        _lib.require_cuda(x_a, x_v)
        want = (self.latent_channels, self.t_v, self.s_v)
        if tuple(x_a.shape[1:]) != want or tuple(x_v.shape[1:]) != want or x_a.shape[0] != x_v.shape[0]:
            raise ValueError(f"av_fusion_forward expects two [B, {want[0]}, {want[1]}, {want[2]}] encodings, got "
                             f"{tuple(x_a.shape)} and {tuple(x_v.shape)}")
        pd = dict(self.named_parameters())
        return _FusionFunction.apply(self, x_a, x_v, *[pd[n] for n in _FUSION_PARAMS])

    # ---- engine ---------------------------------------------------------------------------------------
    def _vis(self, i):
        return self.visual_encoder[4 * i], self.visual_encoder[4 * i + 1]

    def _aud(self, i):
        return self.stft_encoder[3 * i], self.stft_encoder[3 * i + 1]

    # ---- STFT autoencoder engine (K10 + K11) ---------------------------------------------------------------
    def _dec(self, j):
        """(ConvTranspose2d, BatchNorm2d or None) of decoder layer j."""
        idx = 3 * j
        has_bn = self._dec_plan[j][5]
        return self.stft_decoder[idx], (self.stft_decoder[idx + 1] if has_bn else None)

    @staticmethod
    def _pad_c(t, dim, c):
        """zero-pad dimension `dim` of t to c entries (BatchNorm kernels take C = power of two >= 4: a 2-channel layer
        runs with two dead channels, which stay exactly zero through BN (beta 0), tanh and the backward pass)."""
        if t.shape[dim] == c:
            return t.contiguous()
        shp = list(t.shape)
        shp[dim] = c - t.shape[dim]
        return torch.cat((t, torch.zeros(shp, device=t.device, dtype=t.dtype)), dim=dim).contiguous()

    def _bn2d_stats(self, y, bn, c_real, count, train):
        """(mean, invstd) of a channels-last map whose last c - c_real channels are zero padding."""
        c = y.shape[-1]
        if c == c_real:
            rm, rv = bn.running_mean, bn.running_var
        else:
            rm = self._pad_c(bn.running_mean, 0, c)
            rv = torch.cat((bn.running_var, torch.ones(c - c_real, device=y.device, dtype=torch.float32)))
        if not train:
            return ops.bn_eval_stats(rm, rv, bn.eps)
        mean, invstd = ops.bn_finalize(ops.bn_stats(y, c), count, rm, rv, bn.num_batches_tracked, bn.eps, bn.momentum)
        if c != c_real:
            bn.running_mean.copy_(rm[:c_real])
            bn.running_var.copy_(rv[:c_real])
        return mean, invstd

    def _ae_forward(self, x_a, train=True):
        b = x_a.shape[0]
        assert tuple(x_a.shape[1:]) == (2, self.t_a, self.n_bins)
        x_a = x_a.contiguous().float()
        sv = {"enc": [], "dec": []}
        cur, nchw = x_a, True
        for i, (ci, co, st, pw) in enumerate(self._enc_plan):
            conv, bn = self._aud(i)
            y = ops.conv2d_fwd(cur, conv.weight.detach(), st, pw, nchw)
            ho, wo = y.shape[1], y.shape[2]
            mean, invstd = self._bn2d_stats(y, bn, co, b * ho * wo, train)
            y5 = y.view(b, 1, ho, wo, co)
            out, _ = ops.bn_pool_act_fwd(y5, mean, invstd, bn.weight.detach(), bn.bias.detach(), 1, ops.BN_TANH)
            sv["enc"].append(dict(x=cur, nchw=nchw, y=y5, mean=mean, invstd=invstd, out=out, hw=(ho, wo)))
            cur, nchw = out.view(b, ho, wo, co), False
        for j, (ci, co, k, st, op, has_bn) in enumerate(self._dec_plan):
            convt, bn = self._dec(j)
            ci_p = cur.shape[-1]
            co_p = max(co, 4) if has_bn else co
            w = self._pad_c(self._pad_c(convt.weight.detach(), 0, ci_p), 1, co_p)
            y = ops.convt2d_fwd(cur, w, st, op, out_nhwc=has_bn)
            rec = dict(x=cur, w=w, hw_in=(cur.shape[1], cur.shape[2]))
            if has_bn:
                ho, wo = y.shape[1], y.shape[2]
                mean, invstd = self._bn2d_stats(y, bn, co, b * ho * wo, train)
                gamma, beta = self._pad_c(bn.weight.detach(), 0, co_p), self._pad_c(bn.bias.detach(), 0, co_p)
                y5 = y.view(b, 1, ho, wo, co_p)
                out, _ = ops.bn_pool_act_fwd(y5, mean, invstd, gamma, beta, 1, ops.BN_TANH)
                rec.update(y=y5, mean=mean, invstd=invstd, gamma=gamma, out=out, hw=(ho, wo))
                cur = out.view(b, ho, wo, co_p)
            else:
                cur = y
            sv["dec"].append(rec)
        return cur, sv

    def _ae_backward(self, sv, d_out, bn_eval=False):
        grads = {}
        bn_reduce = _bn_eval_reduce if bn_eval else None
        b = d_out.shape[0]
        dcur = d_out                                    # NCHW gradient of the last (BN-less) layer's output
        for j in reversed(range(len(self._dec_plan))):
            ci, co, k, st, op, has_bn = self._dec_plan[j]
            s = sv["dec"][j]
            idx = 3 * j
            if has_bn:
                ho, wo = s["hw"]
                co_p = s["gamma"].shape[0]
                gw = torch.empty(co_p, device=d_out.device, dtype=torch.float32)
                gb = torch.empty(co_p, device=d_out.device, dtype=torch.float32)
                dy = ops.bn_pool_act_bwd(dcur.view(b, 1, ho, wo, co_p), s["out"], None, s["y"], s["mean"], s["invstd"], s["gamma"], 1,
                                         ops.BN_TANH, dgamma=gw, dbeta=gb, reduce_fn=bn_reduce).view(b, ho, wo, co_p)
                grads[f"stft_decoder.{idx + 1}.weight"] = gw[:co].clone()
                grads[f"stft_decoder.{idx + 1}.bias"] = gb[:co].clone()
            else:
                dy = dcur
            dw = ops.convt2d_wgrad(s["x"], dy, s["w"].shape, st, op, out_nhwc=has_bn)
            grads[f"stft_decoder.{idx}.weight"] = dw[:ci, :co].contiguous()
            dcur = ops.convt2d_dgrad(dy, s["w"], s["hw_in"], st, op, out_nhwc=has_bn)
        n_layers = len(self._enc_plan)
        for i in reversed(range(n_layers)):
            ci, co, st, pw = self._enc_plan[i]
            conv, bn = self._aud(i)
            s = sv["enc"][i]
            ho, wo = s["hw"]
            gw = torch.empty(co, device=d_out.device, dtype=torch.float32)
            gb = torch.empty(co, device=d_out.device, dtype=torch.float32)
            dy = ops.bn_pool_act_bwd(dcur.view(b, 1, ho, wo, co), s["out"], None, s["y"], s["mean"], s["invstd"], bn.weight.detach(), 1,
                                     ops.BN_TANH, dgamma=gw, dbeta=gb, reduce_fn=bn_reduce).view(b, ho, wo, co)
            grads[f"stft_encoder.{3 * i + 1}.weight"], grads[f"stft_encoder.{3 * i + 1}.bias"] = gw, gb
            grads[f"stft_encoder.{3 * i}.weight"] = ops.conv2d_wgrad(s["x"], dy, conv.weight.shape, st, pw, s["nchw"])
            if i > 0:
                hin, win = sv["enc"][i - 1]["hw"]
                dcur = ops.conv2d_dgrad(dy, conv.weight.detach(), (hin, win), st, pw)
        return grads

    def _fusion_fwd(self, seq, sv):
        """av_fusion_forward (avse_model_final.py:235-251) on the LSTM sequence buffer seq [B,16,2*T*S]: BiLSTM over the
        16 channel steps, fc1, tanh, fc2, tanh (K11-K13).  The Linear layers are weight-streaming (HBM-bound at
        M = batch) with f32 weights in HBM, so they always use the exact-f32 MFMA: bf16 operand rounding would buy no
        time there and costs accuracy; `precise` only switches the conv3d MFMAs."""
        pr = ops.MODE_F32
        b, l = seq.shape[0], self.latent_channels
        seq2d = seq.view(b * l, seq.shape[2])
        gx = torch.empty(b * l, 2048, device=seq.device, dtype=torch.float32)
        ops.gemm(seq2d, self.lstm.weight_ih_l0.detach(), out=gx[:, :1024], precise=pr, split_k=1)
        ops.gemm(seq2d, self.lstm.weight_ih_l0_reverse.detach(), out=gx[:, 1024:], precise=pr, split_k=1)
        av, hp, gs, cs = ops.lstm_fwd(gx.view(b, l, 2, 4, 256), self.lstm.weight_hh_l0.detach(),
                                      self.lstm.weight_hh_l0_reverse.detach())
        h1 = ops.gemm(av.view(b, l * 512), self.fc1.weight.detach(), act=ops.ACT_TANH, precise=pr)
        fused = ops.gemm(h1, self.fc2.weight.detach(), act=ops.ACT_TANH, precise=pr)
        sv.update(seq=seq, av=av, hp=hp, gs=gs, cs=cs, h1=h1, fused=fused)
        return fused

    def _fusion_bwd(self, sv, dfused, wgrad_gemm):
        """Backward of _fusion_fwd: parameter gradients through `wgrad_gemm(name, dz, x)`; returns the gate gradient dgx
        [B*16, 2048], from which _fusion_dseq forms d(seq) -- kept apart so that the gradient all-reduce of the
        fusion weights can start in between."""
        pr = ops.MODE_F32
        seq, fused, h1, av = sv["seq"], sv["fused"], sv["h1"], sv["av"]
        b, l = seq.shape[0], self.latent_channels
        dz2 = ops.act_bwd(dfused, fused, ops.ACT_TANH)
        wgrad_gemm("fc2.weight", dz2, h1)
        dh1 = ops.gemm(dz2, self.fc2.weight.detach(), trans_b=True, precise=pr)
        dz1 = ops.act_bwd(dh1, h1, ops.ACT_TANH)
        wgrad_gemm("fc1.weight", dz1, av.view(b, l * 512))
        dav = ops.gemm(dz1, self.fc1.weight.detach(), trans_b=True, precise=pr)
        dgx = ops.lstm_bwd(dav.view(b, l, 512), self.lstm.weight_hh_l0.detach(), self.lstm.weight_hh_l0_reverse.detach(),
                           sv["gs"], sv["cs"]).view(b * l, 2048)
        seq2d = seq.view(b * l, seq.shape[2])
        hp2 = sv["hp"].view(b * l, 512)
        wgrad_gemm("lstm.weight_ih_l0", dgx[:, :1024], seq2d)
        wgrad_gemm("lstm.weight_ih_l0_reverse", dgx[:, 1024:], seq2d)
        wgrad_gemm("lstm.weight_hh_l0", dgx[:, :1024], hp2[:, :256])
        wgrad_gemm("lstm.weight_hh_l0_reverse", dgx[:, 1024:], hp2[:, 256:])
        return dgx

    def _fusion_dseq(self, dgx, b):
        pr = ops.MODE_F32
        dseq = ops.gemm(dgx[:, :1024], self.lstm.weight_ih_l0.detach(), trans_b=True, precise=pr)
        ops.gemm(dgx[:, 1024:], self.lstm.weight_ih_l0_reverse.detach(), trans_b=True, out=dseq, beta=1, precise=pr)
        return dseq.view(b, self.latent_channels, -1)

    def set_bn_sync(self, reduce_fn):
        """Global-batch BatchNorm for data-parallel training (EXTENSION; the reference is single-device, where BatchNorm
        sees the whole batch, avse_model_final.py:35,...,103): `reduce_fn(t)` must sum the float64 tensor t in place over
        the data-parallel ranks (trainer.TrainStep(sync_bn=True) installs one); None restores per-rank statistics."""
        self._bn_sync = reduce_fn

    def _bn_train_stats(self, part, count, bn):
        if self._bn_sync is None:
            return ops.bn_finalize(part, count, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.eps, bn.momentum)
        return ops.bn_finalize_synced(part, count, self._bn_sync, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                      bn.eps, bn.momentum)

    def _engine_forward(self, x_a, x_v, train=True):
        # train=False (model.eval()): BatchNorm uses its running statistics and leaves them untouched (forward only)
        pr = ops.MODE_F32 if self.precise_fwd else ops.MODE_F16     # forward conv operands: IEEE half (or exact f32)
        b, t, w = x_v.shape[0], self.t_v, self.width
        assert tuple(x_v.shape[1:]) == (1, t, w, w) and tuple(x_a.shape[1:]) == (2, self.t_a, self.n_bins)
        x_v = x_v.contiguous().float()
        x_a = x_a.contiguous().float()
        dev = x_v.device
        ts = t * self.s_v
        seq = torch.empty(b, self.latent_channels, 2 * ts, device=dev, dtype=torch.float32)
        sv = {"x_v": x_v, "x_a": x_a, "seq": seq, "vis": [], "aud": []}
        # --- visual encoder (K7-K9).  16-bit path: every pooled activation is also written as IEEE half by its producer
        # (the rounding the next conv's MFMA staging would apply, done once), so that conv reads half the bytes and copies;
        # the f32 tensor stays for the backward pass (weight gradient operand, BatchNorm / LeakyReLU backward).
        act_in = x_v.view(b, t, w, w)
        act_in16 = act_inb = None
        # training, 16-bit backward: the producers also write a bf16 copy of every activation whose consumer's weight gradient takes one
        # (ops.WGRAD_X16_SHAPES: that kernel then stages both operands by LDS-DMA; the rounding is the one its staging would apply)
        # MEASURED (profiles/r4_wgrad_x16.txt): the weight-gradient launches take 5-12 % less with a bf16 x, the step does not move (829-831 without,
        # 828-829 clips/s with, same box: the copies cost the producers what the consumers gain) -- off unless MAAVSS_WGRAD_X16=1
        want_b = train and not self.precise_bwd and os.environ.get("MAAVSS_WGRAD_X16", "0") == "1"
        for i in range(5):
            conv, bn = self._vis(i)
            co, pad, pool = conv.out_channels, _VIS_PAD[i], _VIS_POOL[i]
            # 16-bit first layer, training forward: the conv output (the step's largest tensor) is never stored -- pass 1 its
            # BatchNorm sums, pass 2 conv again -> BatchNorm -> pool -> LeakyReLU, and the weight gradient recomputes it per tile
            recompute = i == 0 and train and not self.precise_fwd and pool == 2 and self.c1_recompute
            if recompute:
                y, part = ops.conv3d_c1_stats(act_in, conv.weight.detach(), bn.weight.detach())
            elif i == 0:
                y, part = ops.conv3d_c1_fwd(act_in, conv.weight.detach(), want_stats=train, precise=pr)
            else:
                wt = ops.conv3d_prep(conv.weight.detach(), 0, pr)
                y, part = ops.conv3d_igemm(act_in if act_in16 is None else act_in16, wt, co, pad, pr, want_stats=train)
            hh, ww = y.shape[2], y.shape[3]
            if train:
                mean, invstd = self._bn_train_stats(part, b * t * hh * ww, bn)
            else:
                mean, invstd = ops.bn_eval_stats(bn.running_mean, bn.running_var, bn.eps)
            x_b = act_inb
            act_in16 = act_inb = None
            nxt = (co, self._vis(i + 1)[0].out_channels) if i < 4 else None
            wb = want_b and nxt in ops.WGRAD_X16_SHAPES
            if recompute:
                res = ops.conv3d_c1_bn_pool_act(act_in, conv.weight.detach(), mean, invstd, bn.weight.detach(), bn.bias.detach(), want_bf16=wb)
                out, arg, act_in16 = res[:3]
                act_inb = res[3] if wb else None
                strides = None
            elif i < 4:
                if self.precise_fwd:
                    out, arg = ops.bn_pool_act_fwd(y, mean, invstd, bn.weight.detach(), bn.bias.detach(), pool, ops.BN_LEAKY)
                else:
                    res = ops.bn_pool_act_fwd(y, mean, invstd, bn.weight.detach(), bn.bias.detach(), pool, ops.BN_LEAKY, want16=True, want_bf16=wb)
                    out, arg, act_in16 = res[:3]
                    act_inb = res[3] if wb else None
                strides = None
            else:   # write the [B,16,T,S] block of the LSTM sequence directly (avse_model_final.py:58,239-240)
                strides = (self.latent_channels * 2 * ts, self.s_v, 1, 2 * ts)
                out, arg = ops.bn_pool_act_fwd(y, mean, invstd, bn.weight.detach(), bn.bias.detach(), pool, ops.BN_LEAKY,
                                               out=seq, strides=strides)
            sv["vis"].append(dict(x=act_in, x_bf16=x_b, y=y, mean=mean, invstd=invstd, out=out, arg=arg, strides=strides, recompute=recompute))
            act_in = out
        # --- STFT encoder (K10)
        cur, nchw = x_a, True
        n_layers = len(self._enc_plan)
        seq_aud = seq.view(-1)[ts:]
        aud_strides = (self.latent_channels * 2 * ts, 0, 1, 2 * ts)
        for i, (ci, co, st, pw) in enumerate(self._enc_plan):
            conv, bn = self._aud(i)
            y = ops.conv2d_fwd(cur, conv.weight.detach(), st, pw, nchw)
            ho, wo = y.shape[1], y.shape[2]
            if train:
                mean, invstd = self._bn_train_stats(ops.bn_stats(y, co), b * ho * wo, bn)
            else:
                mean, invstd = ops.bn_eval_stats(bn.running_mean, bn.running_var, bn.eps)
            y5 = y.view(b, 1, ho, wo, co)
            last = i == n_layers - 1
            if last and self._enc_pool is None:
                out, _ = ops.bn_pool_act_fwd(y5, mean, invstd, bn.weight.detach(), bn.bias.detach(), 1, ops.BN_TANH,
                                             out=seq_aud, strides=aud_strides)
                strides = aud_strides
            else:
                out, _ = ops.bn_pool_act_fwd(y5, mean, invstd, bn.weight.detach(), bn.bias.detach(), 1, ops.BN_TANH)
                strides = None
            sv["aud"].append(dict(x=cur, nchw=nchw, y=y5, mean=mean, invstd=invstd, out=out, strides=strides, hw=(ho, wo)))
            cur, nchw = out.view(b, ho, wo, co) if strides is None else None, False
        if self._enc_pool is not None:
            ho, wo = sv["aud"][-1]["hw"]
            _lib.call("maavss_adaptive_pool_fwd", cur.data_ptr(), seq_aud.data_ptr(), b, ho, wo, self.latent_channels,
                      self._enc_pool[0], self._enc_pool[1], self.latent_channels * 2 * ts, 1, 2 * ts, _lib.stream_ptr())
        fused = self._fusion_fwd(seq, sv)
        pr = ops.MODE_F32
        # --- heads (K14)
        a = ops.gemm(fused, self.a_fc1[0].weight.detach(), act=ops.ACT_TANH, precise=pr)
        v = ops.gemm(fused, self.v_fc1[0].weight.detach(), act=ops.ACT_SIGMOID, precise=pr)
        sv.update(a=a, v=v)
        a_out = a.view(b, 2, self.output_stft_frames, self.n_bins)
        v_out = v.view(b, self.frame_channels, w, w)
        return (a_out, v_out, fused), sv

    def _engine_backward(self, sv, d_a, d_v, d_fused, need, grads=None, accumulate=False, on_fusion_done=None, on_grad=None,
                         bn_eval=False):
        """Hand-written backward.  `need[name]` says which parameter gradients are wanted; results go into
        `grads[name]` (pre-allocated when `grads` is given -- e.g. views of a flat gradient buffer --,
        `accumulate` then adds instead of overwriting).  `on_fusion_done` is called once the gradients of the
        LSTM / fc / head weights (98 % of the bytes) are complete, so their all-reduce can overlap the rest; `on_grad(name)`
        after each of those weight gradients has been enqueued (trainer.GradSync launches its buckets from it: the heads'
        all-reduce starts while fc2 / fc1 / the LSTM are still in their backward pass)."""
        # conv backward operands: bf16 (gradients need the exponent range), or exact f32; Linear layers always f32
        pr_conv, pr = (ops.MODE_F32 if self.precise_bwd else ops.MODE_BF16), ops.MODE_F32
        bn_reduce = _bn_eval_reduce if bn_eval else self._bn_sync       # `bn_eval`: the forward used running statistics
        out_grads = {}
        pd = dict(self.named_parameters())

        def gbuf(name):
            if grads is not None:
                return grads[name], (1 if accumulate else 0)
            return torch.empty_like(pd[name]), 0

        def wgrad_gemm(name, dz, x):
            """dW[N,K] = dz[M,N]^T @ x[M,K]"""
            if not need.get(name, False):
                return
            buf, beta = gbuf(name)
            ops.gemm(dz, x, trans_a=True, trans_b=True, out=buf, beta=beta, precise=pr)
            out_grads[name] = buf
            if on_grad is not None:
                on_grad(name)

        b, l, t = sv["x_v"].shape[0], self.latent_channels, self.t_v
        ts = t * self.s_v
        a, v, fused = sv["a"], sv["v"], sv["fused"]
        # heads
        dfused = None
        if d_a is not None:
            dz_a = ops.act_bwd(d_a.contiguous().view(b, -1), a, ops.ACT_TANH)
            wgrad_gemm("a_fc1.0.weight", dz_a, fused)
            dfused = ops.gemm(dz_a, self.a_fc1[0].weight.detach(), trans_b=True, precise=pr)
        if d_v is not None:
            dz_v = ops.act_bwd(d_v.contiguous().view(b, -1), v, ops.ACT_SIGMOID)
            wgrad_gemm("v_fc1.0.weight", dz_v, fused)
            if dfused is None:
                dfused = ops.gemm(dz_v, self.v_fc1[0].weight.detach(), trans_b=True, precise=pr)
            else:
                ops.gemm(dz_v, self.v_fc1[0].weight.detach(), trans_b=True, out=dfused, beta=1, precise=pr)
        if d_fused is not None:
            dfused = d_fused.contiguous().clone() if dfused is None else dfused.add_(d_fused)  # tiny [B,512] glue
        if dfused is None:
            raise _lib.MaavssError("backward called without any output gradient")
        # fc2, fc1, LSTM
        dgx = self._fusion_bwd(sv, dfused, wgrad_gemm)
        if on_fusion_done is not None:
            on_fusion_done()
        enc_need = any(need.get(n, False) for n in self._param_names
                       if n.startswith("visual_encoder.") or n.startswith("stft_encoder."))
        if not enc_need:
            return out_grads
        dseq = self._fusion_dseq(dgx, b)

        def bn_grads(prefix, idx):
            nw, nb = f"{prefix}.{idx}.weight", f"{prefix}.{idx}.bias"
            if not (need.get(nw, False) or need.get(nb, False)):
                return None, None, False
            gw, beta = gbuf(nw)
            gb, _ = gbuf(nb)
            out_grads[nw], out_grads[nb] = gw, gb
            return gw, gb, bool(beta)

        # --- STFT encoder backward
        dcur = None
        n_layers = len(self._enc_plan)
        seq_aud_grad = dseq.view(-1)[ts:]
        for i in reversed(range(n_layers)):
            ci, co, st, pw = self._enc_plan[i]
            conv, bn = self._aud(i)
            s = sv["aud"][i]
            ho, wo = s["hw"]
            if i == n_layers - 1:
                if self._enc_pool is not None:
                    dcur = torch.empty(b, ho, wo, co, device=dseq.device, dtype=torch.float32)
                    _lib.call("maavss_adaptive_pool_bwd", seq_aud_grad.data_ptr(), dcur.data_ptr(), b, ho, wo, co,
                              self._enc_pool[0], self._enc_pool[1], l * 2 * ts, 1, 2 * ts, _lib.stream_ptr())
                    dout, out, strides = dcur.view(b, 1, ho, wo, co), s["out"], None
                else:
                    dout, out, strides = seq_aud_grad, sv["seq"].view(-1)[ts:], s["strides"]
            else:
                dout, out, strides = dcur.view(b, 1, ho, wo, co), s["out"], None
            gw, gb, acc = bn_grads("stft_encoder", 3 * i + 1)
            dy = ops.bn_pool_act_bwd(dout, out, None, s["y"], s["mean"], s["invstd"], bn.weight.detach(), 1, ops.BN_TANH,
                                     strides=strides, dgamma=gw, dbeta=gb, accumulate=acc, reduce_fn=bn_reduce).view(b, ho, wo, co)
            wname = f"stft_encoder.{3 * i}.weight"
            if need.get(wname, False):
                buf, beta = gbuf(wname)
                ops.conv2d_wgrad(s["x"], dy, conv.weight.shape, st, pw, s["nchw"], dw=buf, beta=beta)
                out_grads[wname] = buf
            if i > 0:
                hin, win = sv["aud"][i - 1]["hw"]
                dcur = ops.conv2d_dgrad(dy, conv.weight.detach(), (hin, win), st, pw)
        # --- visual encoder backward
        dcur = None
        for i in reversed(range(5)):
            conv, bn = self._vis(i)
            s = sv["vis"][i]
            pad, pool = _VIS_PAD[i], _VIS_POOL[i]
            if i == 4:
                dout, out = dseq, sv["seq"]
            else:
                dout, out = dcur, s["out"]
            gw, gb, acc = bn_grads("visual_encoder", 4 * i + 1)
            wname = f"visual_encoder.{4 * i}.weight"
            if i == 0:
                # first layer: the network input needs no gradient, so dy has one consumer, the weight gradient -- which
                # forms it in its loader from the pooled gradient (no 1.6 GB dy tensor, no dx pass)
                coef = ops.bn_pool_act_bwd(dout, out, s["arg"], s["y"], s["mean"], s["invstd"], bn.weight.detach(), pool,
                                           ops.BN_LEAKY, strides=s["strides"], dgamma=gw, dbeta=gb, accumulate=acc, beta=bn.bias.detach(), coef_only=True,
                                           reduce_fn=bn_reduce)
                if need.get(wname, False):
                    buf, beta = gbuf(wname)
                    if s["recompute"]:
                        ops.conv3d_c1_wgrad_bn_recompute(s["x"], conv.weight.detach(), dout.contiguous(), s["arg"], s["mean"],
                                                         s["invstd"], bn.bias.detach(), coef, pool, dw=buf, beta=beta)
                    else:
                        ops.conv3d_c1_wgrad_bn(s["x"], s["y"], dout.contiguous(), out, s["arg"], s["mean"], s["invstd"], coef, pool,
                                               dw=buf, beta=beta, precise=pr_conv)
                    out_grads[wname] = buf
                continue
            # 16-bit path: dy is written as bf16 -- what both of its consumers (weight gradient, input gradient) round it to
            dy = ops.bn_pool_act_bwd(dout, out, s["arg"], s["y"], s["mean"], s["invstd"], bn.weight.detach(), pool,
                                     ops.BN_LEAKY, strides=s["strides"], dgamma=gw, dbeta=gb, accumulate=acc, beta=bn.bias.detach(),
                                     reduce_fn=bn_reduce, dy_bf16=not self.precise_bwd)
            if need.get(wname, False):
                buf, beta = gbuf(wname)
                if i == 0:
                    ops.conv3d_c1_wgrad(s["x"], dy, dw=buf, beta=beta)
                else:
                    xw = s["x_bf16"] if s.get("x_bf16") is not None and dy.dtype == torch.bfloat16 else s["x"]
                    ops.conv3d_wgrad(xw, dy, pad, pr_conv, dw=buf, beta=beta)
                out_grads[wname] = buf
            if i > 0:
                wtd = ops.conv3d_prep(conv.weight.detach(), 1, pr_conv)
                dcur, _ = ops.conv3d_igemm(dy, wtd, conv.in_channels, 4 - pad, pr_conv)
        return out_grads
