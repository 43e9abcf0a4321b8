"""Drop-in for the reference's phasegram variant `avse_model.AV_Fusion_Model` (avse_model.py:410-711; built by
train_av_net.py:66-69) -- SURVEY.md 8 row f1 -- with forward and backward in HIP behind the C-ABI.

Same constructor (`stft_shape, pgram_shape, alpha, latent_channels=64, fc_size=4096`), same `forward(x_a, x_v) ->
(x_a_out, x_v_out, x_av_fused)`, same `visual_ae_forward` / `audio_ae_forward`, same `toggle_*` helpers, same
state_dict keys and shapes (checked against the reference class in oracle/make_golden.py).  The nn modules below are
parameter holders; their forward is never called.

Engine: every Conv2d / ConvTranspose2d (kernels (1,9) and (5,5), with bias) runs through the generic strided
convolution kernels K19 (conv2d_gen.hip), BatchNorm2d + Tanh through the fused BN kernels (channels padded to >= 4
with dead channels that stay exactly zero), the BiLSTM over the phasegram rows through the LSTM step kernels, the
Linear layers through the exact-f32 MFMA GEMM with a fused bias + LeakyReLU(0.3) pass.  Layer plans are computed
analytically; the reference finds them by pushing dummy tensors through the layers as it creates them.
"""
import torch
import torch.nn as nn

from . import _lib, ops
from .avse import _bn_eval_reduce

FUSED_DIM, LSTM_HIDDEN, SLOPE = 512, 256, 0.3


def _plan_pgram_encoder(h, w, latent, fc_size):
    plan, c_in = [], 1
    while w * h * latent > fc_size // 2:            # avse_model.py:431
        if len(plan) > 32:
            raise ValueError("phasegram encoder does not converge")
        c_out = min(c_in * 2, latent)
        plan.append((c_in, c_out))
        w = (w + 8 - 9) // 2 + 1
        c_in = c_out
    return plan, w


def _plan_pgram_decoder(w_enc, w_full, latent):
    plan, c_in, w = [], latent, w_enc
    while w < w_full:                                # avse_model.py:449
        c_out = max(c_in // 2, 1)
        w = 2 * w
        plan.append((c_in, c_out, w != w_full))
        c_in = c_out
    return plan


def _plan_stft_encoder(t_a, n_bins, h, w_enc, latent):
    plan, c_in, cur = [], 2, [t_a, n_bins]
    while cur != [h, w_enc]:                         # avse_model.py:480
        if len(plan) > 32:
            raise ValueError("STFT encoder cannot reach the phasegram code's shape "
                             f"[{h}, {w_enc}] from [{t_a}, {n_bins}] by halving")
        c_out = min(c_in * 4, latent)
        stride = [1, 1]
        for d, tgt in ((0, h), (1, w_enc)):
            if cur[d] > tgt:
                stride[d] = 2
                cur[d] //= 2
        plan.append((c_in, c_out, tuple(stride)))
        c_in = c_out
    return plan


def _plan_stft_decoder(t_a, n_bins, h, w_enc, latent, c_stft):
    plan, c_in, cur = [], latent, [h, w_enc]
    while cur != [t_a, n_bins]:                      # avse_model.py:575
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


def _cpad(c):
    """channel count the BatchNorm kernels accept (power of two >= 4) -- all counts here are powers of two already."""
    return max(c, 4)


class _Fn(torch.autograd.Function):
    """One autograd node per entry point (forward / visual_ae_forward / audio_ae_forward)."""

    @staticmethod
    def forward(ctx, model, mode, *inputs_and_params):
        n_in = 2 if mode == "full" else 1
        outs, saved = model._run_forward(mode, inputs_and_params[:n_in])
        ctx.model, ctx.mode, ctx.saved, ctx.n_in, ctx.was_training = model, mode, saved, n_in, model.training
        return outs

    @staticmethod
    def backward(ctx, *d_outs):
        model = ctx.model
        names = model._names[ctx.mode]
        need = {n: ctx.needs_input_grad[2 + ctx.n_in + i] for i, n in enumerate(names)}
        model._bn_eval = not ctx.was_training       # eval-mode forward: BatchNorm backward without the batch-mean terms
        try:
            grads = model._run_backward(ctx.mode, ctx.saved, d_outs, need)
        finally:
            model._bn_eval = False
        ctx.saved = None
        flat = getattr(model, "_maavss_flat", None)
        if flat is not None:      # FusedAdam steps only parameters that received a gradient (torch.optim.Adam semantics)
            flat.mark(n for n in names if grads.get(n) is not None)
        return (None, None) + (None,) * ctx.n_in + tuple(grads.get(n) for n in names)


_FUSION_PARAMS = ("lstm.weight_ih_l0", "lstm.weight_hh_l0", "lstm.weight_ih_l0_reverse", "lstm.weight_hh_l0_reverse",
                  "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")


class _FusionFn(torch.autograd.Function):
    """av_fusion_forward (avse_model.py:658-670) from given encodings, as one autograd node."""

    @staticmethod
    def forward(ctx, model, x_a, x_v, *params):
        b, h, w = x_a.shape[0], model.h, model.w_enc
        cv, ca = model.c_v, model.c_a
        # permute(0, 2, 1, 3) + cat(dim=2) + flatten == two strided device copies into the LSTM sequence buffer
        seq = torch.empty(b, h, (cv + ca) * w, device=x_a.device, dtype=torch.float32)
        seq[:, :, :cv * w].view(b, h, cv, w).copy_(x_v.permute(0, 2, 1, 3))
        seq[:, :, cv * w:].view(b, h, ca, w).copy_(x_a.permute(0, 2, 1, 3))
        sv = {}
        fused = model._fusion_fwd(seq, sv)
        ctx.model, ctx.saved = model, sv
        return fused

    @staticmethod
    def backward(ctx, d_fused):
        model, sv = ctx.model, ctx.saved
        need = {n: ctx.needs_input_grad[3 + i] for i, n in enumerate(_FUSION_PARAMS)}
        grads = {}
        dgx = model._fusion_bwd(sv, d_fused.contiguous().float(), need, grads)
        d_a = d_v = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            b, h, w, cv, ca = sv["seq"].shape[0], model.h, model.w_enc, model.c_v, model.c_a
            dseq = model._fusion_dseq(dgx).view(b, h, (cv + ca) * w)
            if ctx.needs_input_grad[2]:
                d_v = dseq[:, :, :cv * w].reshape(b, h, cv, w).permute(0, 2, 1, 3)
            if ctx.needs_input_grad[1]:
                d_a = dseq[:, :, cv * w:].reshape(b, h, ca, w).permute(0, 2, 1, 3)
        ctx.saved = None
        flat = getattr(model, "_maavss_flat", None)
        if flat is not None:
            flat.mark(n for n in _FUSION_PARAMS if grads.get(n) is not None)
        return (None, d_a, d_v) + tuple(grads.get(n) for n in _FUSION_PARAMS)


class AV_Fusion_Model(nn.Module):
    def __init__(self, stft_shape, pgram_shape, alpha, latent_channels=64, fc_size=4096):
        super().__init__()
        self.stft_shape, self.pgram_shape, self.latent_channels = list(stft_shape), list(pgram_shape), latent_channels
        self.t_a, self.n_bins = stft_shape[-2], stft_shape[-1]
        self.h, self.w_full = pgram_shape[-2], pgram_shape[-1]
        if stft_shape[1] != 2 or pgram_shape[1] != 1:
            raise ValueError("expected stft_shape [B,2,T_a,F] and pgram_shape [B,1,T,P*P]")
        pe, self.w_enc = _plan_pgram_encoder(self.h, self.w_full, latent_channels, fc_size)
        pd = _plan_pgram_decoder(self.w_enc, self.w_full, latent_channels)
        se = _plan_stft_encoder(self.t_a, self.n_bins, self.h, self.w_enc, latent_channels)
        sd = _plan_stft_decoder(self.t_a, self.n_bins, self.h, self.w_enc, latent_channels, stft_shape[1])
        if not pe or not se:
            raise ValueError("degenerate configuration: an encoder has no layers")
        self.c_v, self.c_a = pe[-1][1], se[-1][1]
        if self.h * 2 * LSTM_HIDDEN != fc_size:
            raise ValueError(f"fc1 expects fc_size = {fc_size} inputs but the BiLSTM over the {self.h} phasegram rows yields "
                             f"{self.h * 2 * LSTM_HIDDEN}: the reference's constructor fails for this shape (avse_model.py:550-553)")

        def seq(layers):
            return nn.Sequential(*[m for group in layers for m in group])

        self.phasegram_encoder = seq([(nn.Conv2d(ci, co, (1, 9), (1, 2), (0, 4)), nn.BatchNorm2d(co), nn.Tanh()) for ci, co in pe])
        self.phasegram_decoder = seq([(nn.ConvTranspose2d(ci, co, (1, 9), (1, 2), (0, 4), (0, 1)),) +
                                      ((nn.BatchNorm2d(co), nn.Tanh()) if bn else ()) for ci, co, bn in pd])
        self.stft_encoder = seq([(nn.Conv2d(ci, co, (5, 5), st, (2, 2)), nn.BatchNorm2d(co), nn.Tanh()) for ci, co, st in se])
        self.lstm = nn.LSTM(input_size=(self.c_v + self.c_a) * self.w_enc, hidden_size=LSTM_HIDDEN, num_layers=1, bias=False,
                            batch_first=True, bidirectional=True)
        self.fc1 = nn.Linear(fc_size, fc_size // 2)
        self.fc2 = nn.Linear(fc_size // 2, FUSED_DIM)
        self.stft_decoder = seq([(nn.ConvTranspose2d(ci, co, (5, 5), st, (2, 2), op),) +
                                 ((nn.BatchNorm2d(co), nn.Tanh()) if bn else ()) for ci, co, st, op, bn in sd])
        self.stft_autoencoder = nn.Sequential(*self.stft_encoder, *self.stft_decoder)
        self.phasegram_autoencoder = nn.Sequential(*self.phasegram_encoder, *self.phasegram_decoder)
        self.a_fc1 = nn.Sequential(nn.Linear(FUSED_DIM, 2 * self.t_a * self.n_bins), nn.LeakyReLU(negative_slope=SLOPE))
        self.v_fc1 = nn.Sequential(nn.Linear(FUSED_DIM, self.h * self.w_full), nn.LeakyReLU(negative_slope=SLOPE))

        # engine descriptors: (prefix, index of the conv module in its Sequential, kind, stride, pad, has_bn)
        def stack(prefix, mods, kind, strides, pads, bns):
            out, idx = [], 0
            for st, pd_, bn in zip(strides, pads, bns):
                out.append(dict(prefix=prefix, idx=idx, kind=kind, stride=st, pad=pd_, bn=bn))
                idx += 3 if bn else 1
            assert idx == len(mods)
            return out

        self._stacks = {
            "pgram_enc": stack("phasegram_encoder", self.phasegram_encoder, "conv", [(1, 2)] * len(pe), [(0, 4)] * len(pe), [True] * len(pe)),
            "pgram_dec": stack("phasegram_decoder", self.phasegram_decoder, "convT", [(1, 2)] * len(pd), [(0, 4)] * len(pd), [p[2] for p in pd]),
            "stft_enc": stack("stft_encoder", self.stft_encoder, "conv", [p[2] for p in se], [(2, 2)] * len(se), [True] * len(se)),
            "stft_dec": stack("stft_decoder", self.stft_decoder, "convT", [p[2] for p in sd], [(2, 2)] * len(sd), [p[4] for p in sd]),
        }
        all_names = [n for n, _ in self.named_parameters()]      # shared modules are listed once, under their first name
        self._names = {
            "full": [n for n in all_names if not n.startswith(("phasegram_decoder.", "stft_decoder."))],
            "visual_ae": [n for n in all_names if n.startswith(("phasegram_encoder.", "phasegram_decoder."))],
            "audio_ae": [n for n in all_names if n.startswith(("stft_encoder.", "stft_decoder."))],
        }

    # ---- reference API: gradient toggles (avse_model.py:624-655) ----------------------------------------------
    def toggle_fusion_grads(self, toggle):
        for m in (self.lstm, self.fc1, self.fc2, self.a_fc1, self.v_fc1):
            m.requires_grad_(toggle)

    def toggle_stft_ae_grads(self, toggle):
        for m in self.stft_autoencoder:
            m.requires_grad_(toggle)

    def toggle_phasegram_ae_grads(self, toggle):
        for m in self.phasegram_autoencoder:
            m.requires_grad_(toggle)

    def toggle_enc_grads(self, toggle):
        for m in list(self.stft_encoder) + list(self.phasegram_encoder):
            m.requires_grad_(toggle)

    def toggle_dec_grads(self, toggle):
        for m in list(self.stft_decoder) + list(self.phasegram_decoder):
            m.requires_grad_(toggle)

    # ---- entry points ----------------------------------------------------------------------------------------
    def _engine(self, mode, *inputs):
        _lib.require_cuda(*inputs)
        pd = dict(self.named_parameters())
        return _Fn.apply(self, mode, *inputs, *[pd[n] for n in self._names[mode]])

    def forward(self, x_a, x_v):
        return self._engine("full", x_a, x_v)

    def visual_ae_forward(self, x_v):
        return self._engine("visual_ae", x_v)

    def audio_ae_forward(self, x_a):
        return self._engine("audio_ae", x_a)

    def av_fusion_forward(self, x_a, x_v):
        """avse_model.py:658-670: encodings x_a [B, c_a, h, w], x_v [B, c_v, h, w] -> x_av_fused [B, 512] (permute to rows,
        cat on the channel dim, flatten, BiLSTM over the h rows, fc1, LeakyReLU(0.3), fc2, LeakyReLU(0.3)) as one autograd
        node over the HIP engine; differentiable in both encodings and the LSTM / fc weights and biases."""
        _lib.require_cuda(x_a, x_v)
        want_a, want_v = (self.c_a, self.h, self.w_enc), (self.c_v, self.h, self.w_enc)
        if tuple(x_a.shape[1:]) != want_a or tuple(x_v.shape[1:]) != want_v or x_a.shape[0] != x_v.shape[0]:
            raise ValueError(f"av_fusion_forward expects encodings [B, {want_a[0]}, {want_a[1]}, {want_a[2]}] (audio) and "
                             f"[B, {want_v[0]}, {want_v[1]}, {want_v[2]}] (phasegram), got {tuple(x_a.shape)} and {tuple(x_v.shape)}")
        pd = dict(self.named_parameters())
        return _FusionFn.apply(self, x_a, x_v, *[pd[n] for n in _FUSION_PARAMS])

    # ---- engine: convolution stacks ---------------------------------------------------------------------------
    def _mods(self, layer):
        seq = getattr(self, layer["prefix"])
        conv = seq[layer["idx"]]
        return conv, (seq[layer["idx"] + 1] if layer["bn"] else None)

    def _bn_stats(self, y, bn, c, count, train):
        cp = y.shape[-1]
        if cp == c:
            rm, rv = bn.running_mean, bn.running_var
        else:
            z = torch.zeros(cp - c, device=y.device, dtype=torch.float32)
            rm, rv = torch.cat((bn.running_mean, z)), torch.cat((bn.running_var, z + 1))
        if not train:
            return ops.bn_eval_stats(rm, rv, bn.eps)
        mean, invstd = ops.bn_finalize(ops.bn_stats(y, cp), count, rm, rv, bn.num_batches_tracked, bn.eps, bn.momentum)
        if cp != c:
            bn.running_mean.copy_(rm[:c])
            bn.running_var.copy_(rv[:c])
        return mean, invstd

    @staticmethod
    def _pad1(t, cp):
        return t if t.shape[0] == cp else torch.cat((t, torch.zeros(cp - t.shape[0], device=t.device, dtype=t.dtype)))

    def _stack_forward(self, name, x_map, train, final_nchw=False, seq_out=None):
        """x_map: ops.Map of the input.  Runs conv(+bias) -> BN -> tanh per layer.  `seq_out` = (buffer, element offset,
        strides) makes the last layer's BN+tanh write straight into the LSTM sequence buffer.  Returns (output, saved)."""
        saved, cur = [], x_map
        layers = self._stacks[name]
        b = x_map.b
        for li, layer in enumerate(layers):
            conv, bn = self._mods(layer)
            w, bias = conv.weight.detach(), conv.bias.detach()
            st, pd_ = layer["stride"], layer["pad"]
            kh, kw = w.shape[2], w.shape[3]
            last = li == len(layers) - 1
            if layer["kind"] == "conv":
                co = w.shape[0]
                ho, wo = (cur.h + 2 * pd_[0] - kh) // st[0] + 1, (cur.w + 2 * pd_[1] - kw) // st[1] + 1
            else:
                co = w.shape[1]
                op = conv.output_padding
                ho, wo = (cur.h - 1) * st[0] - 2 * pd_[0] + kh + op[0], (cur.w - 1) * st[1] - 2 * pd_[1] + kw + op[1]
            if bn is None and final_nchw:
                y = torch.empty(b, co, ho, wo, device=cur.t.device, dtype=torch.float32)
                ymap = ops.Map(y, nchw=True)
            else:
                cp = _cpad(co) if bn is not None else co
                y = torch.zeros(b, ho, wo, cp, device=cur.t.device, dtype=torch.float32) if cp != co else \
                    torch.empty(b, ho, wo, cp, device=cur.t.device, dtype=torch.float32)
                ymap = ops.Map(y, c=co)
            if layer["kind"] == "conv":
                ops.conv_gen_small(cur, w, bias, ymap, st, pd_)
            else:
                ops.conv_gen_big(cur, w, bias, ymap, st, pd_)
            rec = dict(x=cur, y=ymap)
            if bn is not None:
                cp = y.shape[-1]
                mean, invstd = self._bn_stats(y, bn, co, b * ho * wo, train)
                gamma, beta = self._pad1(bn.weight.detach(), cp), self._pad1(bn.bias.detach(), cp)
                y5 = y.view(b, ho, 1, wo, cp)                   # "T" = rows, one spatial row each: strides per row / column
                if last and seq_out is not None:
                    buf, off, strides = seq_out
                    assert cp == co
                    out, _ = ops.bn_pool_act_fwd(y5, mean, invstd, gamma, beta, 1, ops.BN_TANH, out=buf.view(-1)[off:], strides=strides)
                    rec.update(y5=y5, mean=mean, invstd=invstd, gamma=gamma, out=out, strides=strides)
                    cur = None
                else:
                    out, _ = ops.bn_pool_act_fwd(y5, mean, invstd, gamma, beta, 1, ops.BN_TANH)
                    rec.update(y5=y5, mean=mean, invstd=invstd, gamma=gamma, out=out, strides=None)
                    cur = ops.Map(out.view(b, ho, wo, cp), c=co)
            else:
                cur = ymap
            saved.append(rec)
        return cur, saved

    def _stack_backward(self, name, saved, dcur, need, grads, want_dx=False, dcur_strided=None):
        """dcur: gradient w.r.t. the stack output -- a tensor shaped like the last layer's output (NHWC padded / NCHW), or
        `dcur_strided` = (flat buffer view, out view) for an output that lives in the LSTM sequence buffer."""
        layers = self._stacks[name]
        dx = None
        for li in reversed(range(len(layers))):
            layer, s = layers[li], saved[li]
            conv, bn = self._mods(layer)
            prefix, idx = layer["prefix"], layer["idx"]
            w = conv.weight.detach()
            st, pd_ = layer["stride"], layer["pad"]
            ymap = s["y"]
            b = ymap.b
            if bn is not None:
                cp = s["y5"].shape[-1]
                gw = torch.empty(cp, device=w.device, dtype=torch.float32)
                gb = torch.empty(cp, device=w.device, dtype=torch.float32)
                if s["strides"] is not None:
                    dout, out = dcur_strided
                else:
                    dout, out = dcur.view(s["y5"].shape), s["out"]
                dy = ops.bn_pool_act_bwd(dout, out, None, s["y5"], s["mean"], s["invstd"], s["gamma"], 1, ops.BN_TANH,
                                         strides=s["strides"], dgamma=gw, dbeta=gb,
                                         reduce_fn=_bn_eval_reduce if getattr(self, "_bn_eval", False) else None)
                if need.get(f"{prefix}.{idx + 1}.weight", False):
                    grads[f"{prefix}.{idx + 1}.weight"] = gw[:ymap.c].clone()
                if need.get(f"{prefix}.{idx + 1}.bias", False):
                    grads[f"{prefix}.{idx + 1}.bias"] = gb[:ymap.c].clone()
                dymap = ops.Map(dy.view(b, ymap.h, ymap.w, cp), c=ymap.c)
            else:
                dymap = ops.Map(dcur, nchw=ymap.nchw, c=ymap.c)
            if need.get(f"{prefix}.{idx}.bias", False):
                grads[f"{prefix}.{idx}.bias"] = ops.channel_sum(dymap)
            if need.get(f"{prefix}.{idx}.weight", False):
                if layer["kind"] == "conv":
                    grads[f"{prefix}.{idx}.weight"] = ops.conv_gen_wgrad(dymap, s["x"], w.shape, st, pd_)
                else:
                    grads[f"{prefix}.{idx}.weight"] = ops.conv_gen_wgrad(s["x"], dymap, w.shape, st, pd_)
            if li > 0 or want_dx:
                xm = s["x"]
                dxt = torch.zeros_like(xm.t) if xm.c != xm.c_alloc else torch.empty_like(xm.t)
                dxm = ops.Map(dxt, nchw=xm.nchw, c=xm.c)
                if layer["kind"] == "conv":
                    ops.conv_gen_big(dymap, w, None, dxm, st, pd_)
                else:
                    ops.conv_gen_small(dymap, w, None, dxm, st, pd_)
                dcur, dx = dxt, dxt
        return dx

    # ---- engine: whole network ----------------------------------------------------------------------------------
    def _run_forward(self, mode, inputs):
        train = self.training
        if mode == "visual_ae":
            x_v = inputs[0].contiguous().float()
            assert tuple(x_v.shape[1:]) == (1, self.h, self.w_full)
            code, s_enc = self._stack_forward("pgram_enc", ops.Map(x_v, nchw=True), train)
            out, s_dec = self._stack_forward("pgram_dec", code, train, final_nchw=True)
            return out.t, dict(enc=s_enc, dec=s_dec)
        if mode == "audio_ae":
            x_a = inputs[0].contiguous().float()
            assert tuple(x_a.shape[1:]) == (2, self.t_a, self.n_bins)
            code, s_enc = self._stack_forward("stft_enc", ops.Map(x_a, nchw=True), train)
            out, s_dec = self._stack_forward("stft_dec", code, train, final_nchw=True)
            return out.t, dict(enc=s_enc, dec=s_dec)
        x_a, x_v = inputs[0].contiguous().float(), inputs[1].contiguous().float()
        b = x_a.shape[0]
        assert tuple(x_a.shape[1:]) == (2, self.t_a, self.n_bins) and tuple(x_v.shape[1:]) == (1, self.h, self.w_full)
        dev, h, w = x_a.device, self.h, self.w_enc
        feat = (self.c_v + self.c_a) * w
        seq = torch.empty(b, h, feat, device=dev, dtype=torch.float32)      # cat((x_v, x_a), channel) after permute (0,2,1,3), flattened
        strides = (h * feat, feat, 1, w)                                   # (batch, row, column, channel) of the sequence buffer
        _, s_v = self._stack_forward("pgram_enc", ops.Map(x_v, nchw=True), train, seq_out=(seq, 0, strides))
        _, s_a = self._stack_forward("stft_enc", ops.Map(x_a, nchw=True), train, seq_out=(seq, self.c_v * w, strides))
        pr = ops.MODE_F32
        saved = dict(s_v=s_v, s_a=s_a, strides=strides)
        fused = self._fusion_fwd(seq, saved)
        a = ops.bias_act_(ops.gemm(fused, self.a_fc1[0].weight.detach(), precise=pr), self.a_fc1[0].bias.detach(), ops.ACT_LEAKY, SLOPE)
        v = ops.bias_act_(ops.gemm(fused, self.v_fc1[0].weight.detach(), precise=pr), self.v_fc1[0].bias.detach(), ops.ACT_LEAKY, SLOPE)
        saved.update(a=a, v=v)
        return (a.view(x_a.shape), v.view(x_v.shape), fused), saved

    def _fusion_fwd(self, seq, sv):
        """av_fusion_forward (avse_model.py:658-670) on the sequence buffer seq [B, h, (c_v + c_a) * w]: BiLSTM over the h
        phasegram rows, fc1 + bias + LeakyReLU(0.3), fc2 + bias + LeakyReLU(0.3)."""
        pr = ops.MODE_F32
        b, h, feat = seq.shape
        seq2d = seq.view(b * h, feat)
        gx = torch.empty(b * h, 2048, device=seq.device, dtype=torch.float32)
        ops.gemm(seq2d, self.lstm.weight_ih_l0.detach(), out=gx[:, :1024], precise=pr, split_k=1)
        ops.gemm(seq2d, self.lstm.weight_ih_l0_reverse.detach(), out=gx[:, 1024:], precise=pr, split_k=1)
        av, hp, gs, cs = ops.lstm_fwd(gx.view(b, h, 2, 4, 256), self.lstm.weight_hh_l0.detach(), self.lstm.weight_hh_l0_reverse.detach())
        h1 = ops.bias_act_(ops.gemm(av.view(b, h * 512), self.fc1.weight.detach(), precise=pr), self.fc1.bias.detach(), ops.ACT_LEAKY, SLOPE)
        fused = ops.bias_act_(ops.gemm(h1, self.fc2.weight.detach(), precise=pr), self.fc2.bias.detach(), ops.ACT_LEAKY, SLOPE)
        sv.update(seq=seq, av=av, hp=hp, gs=gs, cs=cs, h1=h1, fused=fused)
        return fused

    def _fusion_bwd(self, sv, dfused, need, grads):
        """Backward of _fusion_fwd: parameter gradients into `grads`; returns the LSTM gate gradient dgx [B*h, 2048]."""
        pr = ops.MODE_F32
        b, h = sv["seq"].shape[0], self.h
        fused, h1, av = sv["fused"], sv["h1"], sv["av"]

        def lin(name_w, name_b, dz, x):
            if need.get(name_w, False):
                grads[name_w] = ops.gemm(dz, x, trans_a=True, trans_b=True, precise=pr)
            if need.get(name_b, False):
                grads[name_b] = ops.rows_sum(dz)

        dz2 = ops.leaky_bwd(dfused, fused, SLOPE)
        lin("fc2.weight", "fc2.bias", dz2, h1)
        dh1 = ops.gemm(dz2, self.fc2.weight.detach(), trans_b=True, precise=pr)
        dz1 = ops.leaky_bwd(dh1, h1, SLOPE)
        lin("fc1.weight", "fc1.bias", dz1, av.view(b, h * 512))
        dav = ops.gemm(dz1, self.fc1.weight.detach(), trans_b=True, precise=pr)
        dgx = ops.lstm_bwd(dav.view(b, h, 512), self.lstm.weight_hh_l0.detach(), self.lstm.weight_hh_l0_reverse.detach(),
                           sv["gs"], sv["cs"]).view(b * h, 2048)
        seq2d, hp2 = sv["seq"].view(b * h, -1), sv["hp"].view(b * h, 512)
        for nm, dz, x in (("lstm.weight_ih_l0", dgx[:, :1024], seq2d), ("lstm.weight_ih_l0_reverse", dgx[:, 1024:], seq2d),
                          ("lstm.weight_hh_l0", dgx[:, :1024], hp2[:, :256]), ("lstm.weight_hh_l0_reverse", dgx[:, 1024:], hp2[:, 256:])):
            if need.get(nm, False):
                grads[nm] = ops.gemm(dz, x, trans_a=True, trans_b=True, precise=pr)
        return dgx

    def _fusion_dseq(self, dgx):
        pr = ops.MODE_F32
        dseq = ops.gemm(dgx[:, :1024], self.lstm.weight_ih_l0.detach(), trans_b=True, precise=pr)
        ops.gemm(dgx[:, 1024:], self.lstm.weight_ih_l0_reverse.detach(), trans_b=True, out=dseq, beta=1, precise=pr)
        return dseq

    def _run_backward(self, mode, sv, d_outs, need):
        grads = {}
        if mode in ("visual_ae", "audio_ae"):
            enc, dec = ("pgram_enc", "pgram_dec") if mode == "visual_ae" else ("stft_enc", "stft_dec")
            d = d_outs[0].contiguous().float()
            enc_need = any(need.get(n, False) for n in self._names[mode] if n.startswith(self._stacks[enc][0]["prefix"]))
            dcode = self._stack_backward(dec, sv["dec"], d, need, grads, want_dx=enc_need)
            if enc_need:
                self._stack_backward(enc, sv["enc"], dcode, need, grads)
            return grads
        d_a, d_v, d_fused = d_outs
        pr = ops.MODE_F32
        b, h, w = sv["seq"].shape[0], self.h, self.w_enc
        a, v, fused, h1, av = sv["a"], sv["v"], sv["fused"], sv["h1"], sv["av"]

        def lin(name_w, name_b, dz, x):
            if need.get(name_w, False):
                grads[name_w] = ops.gemm(dz, x, trans_a=True, trans_b=True, precise=pr)
            if need.get(name_b, False):
                grads[name_b] = ops.rows_sum(dz)

        dfused = None
        if d_a is not None:
            dz_a = ops.leaky_bwd(d_a.contiguous().float().view(b, -1), a, SLOPE)
            lin("a_fc1.0.weight", "a_fc1.0.bias", dz_a, fused)
            dfused = ops.gemm(dz_a, self.a_fc1[0].weight.detach(), trans_b=True, precise=pr)
        if d_v is not None:
            dz_v = ops.leaky_bwd(d_v.contiguous().float().view(b, -1), v, SLOPE)
            lin("v_fc1.0.weight", "v_fc1.0.bias", dz_v, fused)
            if dfused is None:
                dfused = ops.gemm(dz_v, self.v_fc1[0].weight.detach(), trans_b=True, precise=pr)
            else:
                ops.gemm(dz_v, self.v_fc1[0].weight.detach(), trans_b=True, out=dfused, beta=1, precise=pr)
        if d_fused is not None:
            dfused = d_fused.contiguous().clone() if dfused is None else dfused.add_(d_fused)      # tiny [B,512] glue
        if dfused is None:
            raise _lib.MaavssError("backward called without any output gradient")
        dgx = self._fusion_bwd(sv, dfused, need, grads)
        enc_need = any(need.get(n, False) for n in self._names["full"] if n.startswith(("phasegram_encoder.", "stft_encoder.")))
        if not enc_need:
            return grads
        dseq = self._fusion_dseq(dgx)
        flat_d, flat_o = dseq.view(-1), sv["seq"].view(-1)
        self._stack_backward("pgram_enc", sv["s_v"], None, need, grads, dcur_strided=(flat_d, flat_o))
        off = self.c_v * w
        self._stack_backward("stft_enc", sv["s_a"], None, need, grads, dcur_strided=(flat_d[off:], flat_o[off:]))
        return grads
