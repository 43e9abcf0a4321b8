"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the DINO ViT-S/8 attention-frame extractor.

Follows the reference's call sites
  video_attention.py:106-114  vits.vit_small(patch_size=8, num_classes=0), frozen, eval
  video_attention.py:38-57    per frame: crop to a multiple of 8, get_last_selfattention -> [1,6,N+1,N+1],
                              keep the CLS row without its own column: attentions[0,:,0,1:]
  video_attention.py:80-96    reshape [6,h,w], nearest x8 upsample, sum heads, divide by the frame max
  av_dataset.py:323-333       optional temporal diff, divide by the clip max, permute to [1,T,H,W]
(the sort/cumsum/threshold block at video_attention.py:59-78 never reaches the output).

PARITY UNPINNED: `dino/` is an empty un-vendored git submodule of the reference
(.gitmodules:1-3, no pinned commit) and its weights are fetched by URL
(video_attention.py:152-154), unavailable offline.  The network below restates the
published facebookresearch/dino `VisionTransformer` (vit_small: dim 384, depth 12,
6 heads, MLP x4, qkv bias, pre-LN blocks with LayerNorm eps 1e-6, exact-erf GELU,
Conv2d(3,384,k=8,s=8) patch embed, CLS token, learned 785x384 position embedding,
bicubic-interpolated for inputs other than 224^2) with the same state_dict keys;
tests cross-check it against `transformers.ViTModel` built from a local config.
"""
import math
import zlib

import torch
import torch.nn.functional as F

DIM, DEPTH, HEADS, MLP, PATCH = 384, 12, 6, 1536, 8
LN_EPS = 1e-6


def vit_param_shapes(img_size=224):
    n = (img_size // PATCH) ** 2
    sh = {"cls_token": (1, 1, DIM), "pos_embed": (1, n + 1, DIM),
          "patch_embed.proj.weight": (DIM, 3, PATCH, PATCH), "patch_embed.proj.bias": (DIM,),
          "norm.weight": (DIM,), "norm.bias": (DIM,)}
    for i in range(DEPTH):
        p = f"blocks.{i}."
        sh.update({p + "norm1.weight": (DIM,), p + "norm1.bias": (DIM,),
                   p + "attn.qkv.weight": (3 * DIM, DIM), p + "attn.qkv.bias": (3 * DIM,),
                   p + "attn.proj.weight": (DIM, DIM), p + "attn.proj.bias": (DIM,),
                   p + "norm2.weight": (DIM,), p + "norm2.bias": (DIM,),
                   p + "mlp.fc1.weight": (MLP, DIM), p + "mlp.fc1.bias": (MLP,),
                   p + "mlp.fc2.weight": (DIM, MLP), p + "mlp.fc2.bias": (DIM,)})
    return sh


def seeded_vit_state(seed, img_size=224):
    """Random-but-reproducible weights with trained-network-like scales (no DINO checkpoint exists offline)."""
    out = {}
    for k, shape in vit_param_shapes(img_size).items():
        g = torch.Generator(device="cpu")
        g.manual_seed((zlib.crc32(k.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
        r = torch.randn(shape, generator=g)
        if k.endswith("norm1.weight") or k.endswith("norm2.weight") or k == "norm.weight":
            t = 1.0 + 0.1 * r
        elif k.endswith(".bias"):
            t = 0.05 * r
        elif k in ("cls_token", "pos_embed"):
            t = 0.2 * r
        elif k == "patch_embed.proj.weight":
            t = r * (1.0 / math.sqrt(3 * PATCH * PATCH))
        elif "attn.qkv.weight" in k:
            t = r * (1.6 / math.sqrt(DIM))          # sharper-than-init logits so softmax is not flat
        else:
            t = r * (1.0 / math.sqrt(shape[1]))
        out[k] = t.float()
    return out


def interpolate_pos_embed(pos_embed, h_tok, w_tok):
    """Bicubic resize of the patch part of the position embedding (published DINO rule, incl. its +0.1)."""
    n = pos_embed.shape[1] - 1
    if n == h_tok * w_tok and h_tok == w_tok:
        return pos_embed
    side = int(math.sqrt(n))
    patch = pos_embed[:, 1:].reshape(1, side, side, DIM).permute(0, 3, 1, 2)
    patch = F.interpolate(patch, scale_factor=((h_tok + 0.1) / side, (w_tok + 0.1) / side), mode="bicubic")
    assert patch.shape[-2] == h_tok and patch.shape[-1] == w_tok
    patch = patch.permute(0, 2, 3, 1).reshape(1, -1, DIM)
    return torch.cat([pos_embed[:, :1], patch], 1)


# "f16-gelu-half": VideoAttention(gelu="half"); "...-lnpost": VideoAttention(qkv_ln="post") -- norm1 applied after the attn.qkv product
_ROUND = {"bf16": torch.bfloat16, "f16": torch.float16, "f16-gelu-half": torch.float16, "f16-lnpost": torch.float16, "bf16-lnpost": torch.bfloat16}


def _rounder(emulate):
    """emulate: None / False = fp32; "bf16" / "f16" (True = "bf16") = round to that storage format and return f32"""
    if not emulate:
        return lambda t: t
    dt = _ROUND["bf16" if emulate is True else emulate]
    return lambda t: t.to(dt).float()


def prepare_tokens(sd, frames, emulate=None):
    b, _, h, w = frames.shape
    r = _rounder(emulate)
    if emulate:      # vit_patchify stores 16-bit patches, the weight is 16-bit; bias + cls + pos stay f32 (row table)
        x = F.conv2d(r(frames), r(sd["patch_embed.proj.weight"]), None, stride=PATCH) + sd["patch_embed.proj.bias"][None, :, None, None]
    else:
        x = F.conv2d(frames, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=PATCH)
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([sd["cls_token"].expand(b, -1, -1), x], 1)
    return x + interpolate_pos_embed(sd["pos_embed"], h // PATCH, w // PATCH)


QSCALE = 0.125 * 1.4426950408889634      # log2(e) / sqrt(64): the kernels run softmax on exp2
ATT_KT, ATT_THR = 64, 5.0                # key tile and deferred-rescale threshold of maavss_amd/csrc/vit_attn.hip
_GELU_C = (-9.018102001e-10, 7.941707090e-08, -3.038026629e-06, 6.689195681e-05, -9.506666631e-04, 9.298265605e-03,
           -6.552827696e-02, 3.984659427e-01)


def gelu_poly(v):
    """the panel GEMM's GELU (vit_panel_gemm.hip pg_gelu4): v * (1/2 + c Q(c^2)), c = clamp(v, +-4.2), Q a degree-7
    minimax polynomial; |error| vs the exact-erf GELU < 1e-4 -- comparable to the rounding step of small outputs, so the
    emulation has to use the same function."""
    c = v.clamp(-4.2, 4.2)
    u = c * c
    q = u * _GELU_C[0] + _GELU_C[1]
    for k in _GELU_C[2:]:
        q = q * u + k
    return v * (c * q + 0.5)


_GELU_T = (-2.420778323e-01, 4.003976983e-01, -3.264899766e-01, 2.595298127e-01, -2.277023313e-01, 1.849680812e-01,
           -1.484367893e-01, 1.680398153e-01)


def gelu_poly_h2(v):
    """fc1's GELU as VideoAttention(gelu="half") evaluates it (maavss_amd/csrc/vit_epilogue.h pg_gelu_h2, epilogue 4): the same
    polynomial re-expanded in t = c^2 / 16 - 0.55, EVERY operation in IEEE half with one rounding (the packed fused multiply-adds
    are exact up to it: emulated in float64).  Returns f32 holding half-representable values."""
    hd = torch.float16

    def fma(x, y, z):
        return (x.double() * y.double() + z.double()).to(hd)

    def const(x, like):
        return torch.full_like(like, x, dtype=hd)
    h = v.to(hd)
    lim = torch.tensor(4.2, dtype=hd)
    c = torch.minimum(torch.maximum(h, -lim), lim)
    t = fma((c.double() * 0.0625).to(hd), c, const(-0.55, c))
    q = fma(t, const(_GELU_T[0], t), const(_GELU_T[1], t))
    for k in _GELU_T[2:]:
        q = fma(q, t, const(k, t))
    r = fma(c, q, const(0.5, c))
    return (h.double() * r.double()).to(hd).float()


def flash_attention_emulated(q, k, v, r):
    """softmax(q k^T) v of the flash kernel, rounding where it rounds: key tiles of 64 in order, a running maximum that
    only moves when a tile exceeds it by more than 2^5 (first tile: rebased to its own maximum), P = exp2(s - m) rounded
    to the 16-bit format for the P.V product while the row sum keeps the f32 values, O rescaled when m moves.
    q (pre-scaled, log2 units), k, v: [b, heads, n, 64] f32 holding 16-bit-representable values."""
    n = k.shape[-2]
    s_all = q @ k.transpose(-2, -1)
    m = torch.zeros(s_all.shape[:-1] + (1,))
    l = torch.zeros_like(m)
    o = torch.zeros(q.shape)
    for t0 in range(0, n, ATT_KT):
        s = s_all[..., t0:t0 + ATT_KT] - m
        mx = s.max(-1, keepdim=True).values
        delta = torch.where(mx > ATT_THR, mx, torch.zeros_like(mx)) if t0 else mx
        alpha = torch.exp2(-delta)
        m = m + delta
        l, o = l * alpha, o * alpha
        p = torch.exp2(s - delta)
        l = l + p.sum(-1, keepdim=True)
        o = o + r(p) @ v[..., t0:t0 + ATT_KT, :]
    return o / l


def block_forward(sd, i, x, return_attention=False, emulate=None):
    """One pre-LN block.  `emulate` ("bf16" / "f16") rounds exactly where the HIP kernels store 16-bit values:
    the weights, the LayerNorm output, q (after the log2(e)/8 scale) / k / v, the exponentiated probabilities that enter
    P.V (tile-wise, with the kernel's deferred running maximum; the row sum keeps the unrounded f32 values), the attention
    output and the GELU output (the kernel's polynomial GELU for fc1; emulate="f16-gelu-half": evaluated IN half, gelu_poly_h2);
    accumulation, biases, residual stream and softmax stay f32.  What is left between this and the kernels is summation order and hardware exp2 / rsqrt ulps."""
    p = f"blocks.{i}."
    b, n, _ = x.shape
    r = _rounder(emulate)
    if emulate and str(emulate).endswith("-lnpost"):
        # maavss_vit_ws_gemm_ln_post: raw rows rounded, weights W diag(gamma) rounded, row statistics applied behind the product
        w64 = sd[p + "attn.qkv.weight"].double()
        wf = r((w64 * sd[p + "norm1.weight"].double()[None, :]).float())
        mu = x.mean(-1, keepdim=True)
        rstd = (x.var(-1, unbiased=False, keepdim=True) + LN_EPS).rsqrt()
        bprime = (sd[p + "attn.qkv.bias"].double() + w64 @ sd[p + "norm1.bias"].double()).float()
        qkv = rstd * (F.linear(r(x), wf) - mu * wf.sum(-1)) + bprime
    else:
        y = r(F.layer_norm(x, (DIM,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], LN_EPS))
        qkv = F.linear(y, r(sd[p + "attn.qkv.weight"]), sd[p + "attn.qkv.bias"])
    if emulate:
        qkv = r(torch.cat([qkv[..., :DIM] * QSCALE, qkv[..., DIM:]], -1))
    qkv = qkv.reshape(b, n, 3, HEADS, DIM // HEADS).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    if emulate:
        if return_attention:          # vit_cls_attn: exact row maximum, f32 throughout
            s = q @ k.transpose(-2, -1)
            pexp = torch.exp2(s - s.max(-1, keepdim=True).values)
            return pexp / pexp.sum(-1, keepdim=True)
        y = r(flash_attention_emulated(q, k, v, r).transpose(1, 2).reshape(b, n, DIM))
    else:
        att = (q @ k.transpose(-2, -1)) * ((DIM // HEADS) ** -0.5)
        att = att.softmax(-1)
        if return_attention:
            return att
        y = (att @ v).transpose(1, 2).reshape(b, n, DIM)
    x = x + F.linear(y, r(sd[p + "attn.proj.weight"]), sd[p + "attn.proj.bias"])
    y = r(F.layer_norm(x, (DIM,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], LN_EPS))
    h = F.linear(y, r(sd[p + "mlp.fc1.weight"]), sd[p + "mlp.fc1.bias"])
    y = (gelu_poly_h2(h) if emulate == "f16-gelu-half" else r(gelu_poly(h))) if emulate else F.gelu(h)
    return x + F.linear(y, r(sd[p + "mlp.fc2.weight"]), sd[p + "mlp.fc2.bias"])


def get_last_selfattention(sd, frames, return_hidden=False, emulate=None):
    x = prepare_tokens(sd, frames, emulate)
    hidden = [x]
    for i in range(DEPTH - 1):
        x = block_forward(sd, i, x, emulate=emulate)
        hidden.append(x)
    att = block_forward(sd, DEPTH - 1, x, return_attention=True, emulate=emulate)
    return (att, hidden) if return_hidden else att


def cls_attention(sd, frames, emulate=None):
    """[B,3,H,W] -> CLS-row attention without the CLS column, [B, 6, N]."""
    return get_last_selfattention(sd, frames, emulate=emulate)[:, :, 0, 1:]


def attention_frames_from_cls(cls_att, h_tok, w_tok):
    """video_attention.py:80-96 for a stack of frames: [T,6,N] -> [T,1,H,W], each frame /max."""
    t = cls_att.shape[0]
    a = cls_att.reshape(t, HEADS, h_tok, w_tok)
    a = F.interpolate(a, scale_factor=PATCH, mode="nearest")
    a = a.sum(1)
    a = a * (1.0 / a.flatten(1).max(1).values)[:, None, None]
    return a[:, None]


def inference_ref(sd, frames, emulate=None):
    """VideoAttention._inference: frames [T,3,H,W] -> [T,1,H,W] (H, W cropped to multiples of 8 are
    written into a zero canvas of the original size, video_attention.py:39,43-47,96)."""
    t, _, h, w = frames.shape
    hc, wc = h - h % PATCH, w - w % PATCH
    att = cls_attention(sd, frames[:, :, :hc, :wc], emulate)
    out = torch.zeros(t, 1, h, w)
    out[:, :, :hc, :wc] = attention_frames_from_cls(att, hc // PATCH, wc // PATCH)
    return out


def clip_normalise_ref(attn, attn_diff=False):
    """av_dataset.py:323-333: attn [T,1,H,W] -> [1,T,H,W], divided by the clip max."""
    if attn_diff:
        attn = torch.cat([torch.zeros_like(attn[:1]), torch.diff(attn, dim=0)], 0)
    attn = attn * (1.0 / attn.max())
    return attn.permute(1, 0, 2, 3)


def synthetic_frames(n, width, seed):
    """U[0,1) RGB then ImageNet normalisation (av_dataset.py:110-111)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    x = torch.rand(n, 3, width, width, generator=g)
    mean = torch.tensor([0.485, 0.456, 0.406])[None, :, None, None]
    std = torch.tensor([0.229, 0.224, 0.225])[None, :, None, None]
    return (x - mean) / std


# ----------------------------------------------------------------------------------------------------------------
# Block-scaled fp8 (OCP MX) emulation for the config[4] attention path (maavss_amd/csrc/vit_mx.h, vit_attn_mx.hip):
# e4m3 elements, one power-of-two (e8m0) scale per 32 elements along `dim`.  TEST INFRASTRUCTURE.
# ----------------------------------------------------------------------------------------------------------------
def mx_quantise(x, dim):
    """Dequantised values of x after MX quantisation in blocks of 32 along `dim` (size a multiple of 32): scale = the smallest
    power of two s with amax / s <= 448, computed with the kernel's integer formula on amax * (1/448) in float32."""
    x = x.float().movedim(dim, -1)
    shp = x.shape
    xb = x.reshape(*shp[:-1], shp[-1] // 32, 32)
    amax = xb.abs().amax(-1, keepdim=True)
    bits = (amax * torch.tensor(1.0 / 448.0, dtype=torch.float32)).view(torch.int32)
    e = ((bits + 0x7FFFFF) >> 23).clamp(max=253)
    scale = torch.ldexp(torch.ones_like(amax), e - 127)
    q = (xb / scale).to(torch.float8_e4m3fn).float() * scale
    return q.reshape(shp).movedim(-1, dim)


def attention_mx_ref(qkv, frames, ntok, heads=6):
    """softmax(q k^T) v per (frame, head) with the operands quantised where vit_attn_mx_kernel's images are -- q, k per (token, 32 d);
    v per (d, 32 GLOBAL token rows: blocks run over frame boundaries) -- and the flash loop of that kernel: key tiles of 64 starting
    at the 32-aligned global row at or below the frame's first row, a running maximum that moves when a tile exceeds it by more
    than 2^1 (first tile: rebased), P' = 2^7 exp2(s - m) rounded to e4m3 for the P V product AND for the row sum (the very
    weights of O).  qkv [frames * ntok, 3 * heads * 64] with q pre-scaled to log2 units."""
    rows, dim = frames * ntok, heads * 64
    q, k, v = qkv.float().split(dim, 1)
    q8, k8 = mx_quantise(q, 1), mx_quantise(k, 1)
    pad = (-rows) % 32
    vp = torch.cat([v, torch.zeros(pad, dim)]) if pad else v
    v8 = mx_quantise(vp, 0)[:rows]
    out = torch.empty(rows, dim)
    thr, shift = 1.0, 7.0
    for f in range(frames):
        sl = slice(f * ntok, (f + 1) * ntok)
        qf, kf, vf = [t[sl].view(ntok, heads, 64).transpose(0, 1) for t in (q8, k8, v8)]        # [heads, ntok, 64]
        s_all = qf @ kf.transpose(-1, -2)
        lead = (f * ntok) % 32
        m = torch.zeros(heads, ntok, 1)
        l = torch.zeros_like(m)
        o = torch.zeros(heads, ntok, 64)
        t0 = -lead
        first = True
        while t0 < ntok:
            a, b = max(t0, 0), min(t0 + 64, ntok)
            s = s_all[..., a:b] - m
            mx = s.max(-1, keepdim=True).values
            delta = mx if first else torch.where(mx > thr, mx, torch.zeros_like(mx))
            alpha = torch.exp2(-delta)
            m = m + delta
            l, o = l * alpha, o * alpha
            p8 = torch.exp2(s - delta + shift).to(torch.float8_e4m3fn).float()
            l = l + p8.sum(-1, keepdim=True)         # the kernel sums the rounded weights (a ones-row on the matrix pipe)
            o = o + p8 @ vf[..., a:b, :]
            t0 += 64
            first = False
        out[sl] = (o / l).transpose(0, 1).reshape(ntok, dim)
    return out
