"""Thin functional wrappers over the C-ABI (one Python function per entry point family).

Tensors are torch CUDA tensors used purely as device-memory handles: every function passes raw
device pointers + explicit sizes + the current HIP stream to libmaavss_hip.so.  No torch math here.
Layouts: visual activations channels-last [B,T,H,W,C]; STFT-encoder activations NHWC.
"""
import os

import torch

from . import _lib
from ._lib import call, ptr, query, stream_ptr

ACT_NONE, ACT_TANH, ACT_SIGMOID = 0, 1, 2      # gemm / act_bwd
BN_LEAKY, BN_TANH = 0, 1                        # bn_pool_act
MODE_BF16, MODE_F32, MODE_F16 = 0, 1, 2         # MFMA operand arithmetic (`precise` argument of the C-ABI)
BN_EPS, BN_MOMENTUM = 1e-5, 0.1                 # torch.nn.BatchNorm defaults used by the reference


def _f32(*ts):
    for t in ts:
        if t is not None:
            _lib.require_cuda(t)
            assert t.dtype == torch.float32 and t.is_contiguous(), "expected contiguous float32 CUDA tensor"


def gemm(a, b, trans_a=False, trans_b=False, act=ACT_NONE, alpha=1.0, out=None, beta=0, precise=False, split_k=0,
         trans_c=False):
    """C[M,N] = act(alpha * op(a) @ op(b)^T) (+ C).  a: [M,K] (or [K,M] if trans_a); b: [N,K] (or [K,N] if trans_b).
    2-D row-major views with unit inner stride are accepted (row stride = leading dimension)."""
    for t in (a, b):
        _lib.require_cuda(t)
        assert t.dim() == 2 and t.dtype == torch.float32 and t.stride(1) == 1
    m, k = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    n, k2 = (b.shape[1], b.shape[0]) if trans_b else (b.shape[0], b.shape[1])
    assert k == k2, f"gemm: inner dimensions differ ({k} vs {k2})"
    if out is None:
        assert beta == 0
        out = torch.empty((n, m) if trans_c else (m, n), device=a.device, dtype=torch.float32)
    assert out.dim() == 2 and out.stride(1) == 1 and out.dtype == torch.float32
    call("maavss_gemm_f32", ptr(a), a.stride(0), int(trans_a), ptr(b), b.stride(0), int(trans_b), ptr(out),
         out.stride(0), int(trans_c), m, n, k, float(alpha), int(beta), int(act), int(split_k), int(precise),
         stream_ptr())
    return out


# ---------------------------------------------------------------------------------------------- conv3d
def conv3d_prep(w, mode, precise):
    """reference-layout weight [Co,Ci,3,5,5] -> re-laid image for conv3d_igemm (mode 0 fwd, 1 dgrad)."""
    _f32(w)
    co, ci = w.shape[0], w.shape[1]
    cin, n = (co, ci) if mode else (ci, co)
    kp = query("maavss_conv3d_kp", cin)
    wt = torch.empty(3 * n * kp, device=w.device, dtype=torch.float32 if int(precise) == MODE_F32 else torch.int16)
    call("maavss_conv3d_prep_weights", ptr(w), ptr(wt), co, ci, int(mode), int(precise), stream_ptr())
    return wt


def conv3d_igemm(x, wt, c_out, pad, precise, want_stats=False):
    """x [B,T,H,W,Ci] -> y [B,T,Ho,Wo,c_out] (+ BatchNorm partial sums [nblk,2,c_out]).  x is f32, or already in the
    16-bit MFMA operand format of `precise` (float16 for MODE_F16, bfloat16 for MODE_BF16: copied, not converted)."""
    _lib.require_cuda(x)
    x16 = x.dtype != torch.float32
    if x16:
        want = {MODE_F16: torch.float16, MODE_BF16: torch.bfloat16}.get(int(precise))
        assert x.dtype == want and x.is_contiguous(), "16-bit conv3d input must match the MFMA mode"
    else:
        _f32(x)
    b, t, h, w, ci = x.shape
    ho, wo = h + 2 * pad - 4, w + 2 * pad - 4
    y = torch.empty(b, t, ho, wo, c_out, device=x.device, dtype=torch.float32)
    part = None
    if want_stats:
        nblk = ((wo + 15) // 16) * ((ho + 15) // 16) * b * t
        part = torch.empty(nblk, 2, c_out, device=x.device, dtype=torch.float32)
    call("maavss_conv3d_igemm", ptr(x), ptr(wt), ptr(y), ptr(part), b, t, h, w, ci, c_out, pad, int(precise), int(x16),
         stream_ptr())
    return y, part


def wgrad_chunks(b, t, ho, wo, ci=64, co=64):
    """Number of position chunks (= partial sums) of the weight-gradient kernels.  The wide kernel (16->32: one
    workgroup per chunk, 32->64: three) needs more chunks than the 15-workgroups-per-chunk kernel to fill 256 CUs."""
    tiles = b * t * ((ho + 15) // 16) * ((wo + 15) // 16)
    if (ci, co) == (16, 32):
        return max(1, min(512, tiles // 2))
    if (ci, co) == (32, 64):
        return max(1, min(256, tiles // 2))
    if (ci, co) == (64, 64):      # 16-bit path: wide kernel over two 32-channel halves, 3 x 2 workgroups per chunk
        return max(1, min(85, tiles // 4))
    return max(1, min(64, tiles // 4))


WGRAD_X16_SHAPES = ((16, 32), (32, 64), (64, 64))      # (C_in, C_out) whose weight-gradient kernel takes a bf16 x (LDS-DMA staging)


def conv3d_wgrad(x, dy, pad, precise, dw=None, beta=0, nchunk=None):
    """dy: f32, or bfloat16 (bn_pool_act_bwd(dy_bf16=True)) with precise = MODE_BF16; x: f32, or -- with a bfloat16 dy and one of
    WGRAD_X16_SHAPES -- the bfloat16 copy its producer wrote (bn_pool_act_fwd(want_bf16=True)): the same result bit for bit."""
    _f32(dw)
    dy16 = dy.dtype == torch.bfloat16
    if dy16:
        assert int(precise) == MODE_BF16 and dy.is_contiguous() and dy.is_cuda
    else:
        _f32(dy)
    x16 = x.dtype == torch.bfloat16
    if x16:
        assert dy16 and x.is_contiguous() and x.is_cuda and (x.shape[-1], dy.shape[-1]) in WGRAD_X16_SHAPES
    else:
        _f32(x)
    b, t, h, w, ci = x.shape
    co = dy.shape[-1]
    ho, wo = h + 2 * pad - 4, w + 2 * pad - 4
    assert tuple(dy.shape) == (b, t, ho, wo, co)
    if nchunk is None:
        nchunk = wgrad_chunks(b, t, ho, wo, ci, co)
    ws = torch.empty(query("maavss_conv3d_wgrad_ws_bytes", ci, co, nchunk) // 4, device=x.device, dtype=torch.float32)
    if dw is None:
        dw = torch.empty(co, ci, 3, 5, 5, device=x.device, dtype=torch.float32)
        beta = 0
    call("maavss_conv3d_wgrad", ptr(x), ptr(dy), ptr(dw), ptr(ws), nchunk, b, t, h, w, ci, co, pad, int(beta),
         int(precise), int(dy16) | (int(x16) << 1), stream_ptr())
    return dw


def conv3d_c1_fwd(x, w, want_stats=False, precise=MODE_F32):
    """x [B,T,H,W] (C=1), w [16,1,3,5,5] -> y [B,T,H,W,16].  precise: MODE_F32 (exact VALU) or MODE_F16 (MFMA)."""
    _f32(x, w)
    b, t, h, wd = x.shape
    y = torch.empty(b, t, h, wd, 16, device=x.device, dtype=torch.float32)
    w16 = torch.empty(1200, device=x.device, dtype=torch.float32)
    part = None
    if want_stats:
        part = torch.empty(query("maavss_conv3d_c1_fwd_nparts", b, t, h, wd, int(precise)), 2, 16, device=x.device, dtype=torch.float32)
    call("maavss_conv3d_c1_fwd", ptr(x), ptr(w), ptr(w16), ptr(y), ptr(part), b, t, h, wd, int(precise), stream_ptr())
    return y, part


def conv3d_c1_wgrad(x, dy, dw=None, beta=0, nchunk=None):
    _f32(x, dy, dw)
    b, t, h, wd = x.shape
    if nchunk is None:
        nchunk = max(1, min(1024, (b * t * ((h + 15) // 16) * ((wd + 15) // 16)) // 2))
    ws = torch.empty(nchunk * 1200, device=x.device, dtype=torch.float32)
    if dw is None:
        dw = torch.empty(16, 1, 3, 5, 5, device=x.device, dtype=torch.float32)
        beta = 0
    call("maavss_conv3d_c1_wgrad", ptr(x), ptr(dy), ptr(dw), ptr(ws), nchunk, b, t, h, wd, int(beta), stream_ptr())
    return dw


def conv3d_c1_wgrad_bn(x, y, dout, out, arg, mean, invstd, coef, pool, dw=None, beta=0, nchunk=None, precise=MODE_F32):
    """first-layer weight gradient straight from the pooled gradient: BN / max-pool / LeakyReLU backward in the loader."""
    _f32(x, y, dout, out, mean, invstd, coef, dw)
    b, t, h, wd = x.shape
    assert y.shape == (b, t, h, wd, 16) and dout.is_contiguous() and out.is_contiguous()
    if nchunk is None:
        cap = int(os.environ.get("MAAVSS_C1_NCHUNK", "1024"))
        nchunk = max(1, min(cap, (b * t * ((h + 15) // 16) * ((wd + 15) // 16)) // 2))
    ws = torch.empty(nchunk * 1200, device=x.device, dtype=torch.float32)
    if dw is None:
        dw, beta = torch.empty(16, 1, 3, 5, 5, device=x.device, dtype=torch.float32), 0
    call("maavss_conv3d_c1_wgrad_bn", ptr(x), ptr(y), ptr(dout), ptr(out), ptr(arg), ptr(mean), ptr(invstd), ptr(coef), pool, ptr(dw),
         ptr(ws), nchunk, b, t, h, wd, int(beta), int(precise), stream_ptr())
    return dw


def conv3d_c1_stats(x, w, gamma):
    """16-bit first layer, pass 1: BatchNorm partial sums of the conv output, which is NOT stored (the returned y is
    uninitialised memory unless some |gamma| < 1e-2: the degenerate case in which the BatchNorm backward gathers from it)."""
    _f32(x, w, gamma)
    b, t, h, wd = x.shape
    y = torch.empty(b, t, h, wd, 16, device=x.device, dtype=torch.float32)
    part = torch.empty(query("maavss_conv3d_c1_fwd_nparts", b, t, h, wd, MODE_F16), 2, 16, device=x.device, dtype=torch.float32)
    call("maavss_conv3d_c1_stats", ptr(x), ptr(w), ptr(gamma), ptr(y), ptr(part), b, t, h, wd, stream_ptr())
    return y, part


def conv3d_c1_bn_pool_act(x, w, mean, invstd, gamma, beta, want16=True, want_bf16=False):
    """pass 2: conv again -> BatchNorm -> MaxPool(1,2,2) -> LeakyReLU: (out f32, argmax u8, out16 IEEE half[, the bfloat16 copy with
    want_bf16: the next layer's weight-gradient operand]), [B,T,H//2,W//2,16]."""
    _f32(x, w, mean, invstd, gamma, beta)
    b, t, h, wd = x.shape
    hp, wp = h // 2, wd // 2
    out = torch.empty(b, t, hp, wp, 16, device=x.device, dtype=torch.float32)
    arg = torch.empty(b, t, hp, wp, 16, device=x.device, dtype=torch.uint8)
    out16 = torch.empty(b, t, hp, wp, 16, device=x.device, dtype=torch.float16) if want16 else None
    outb = torch.empty(b, t, hp, wp, 16, device=x.device, dtype=torch.bfloat16) if want_bf16 else None
    call("maavss_conv3d_c1_bn_pool_act", ptr(x), ptr(w), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), ptr(out), ptr(out16), ptr(outb),
         ptr(arg), b, t, h, wd, stream_ptr())
    return (out, arg, out16, outb) if want_bf16 else (out, arg, out16)


def conv3d_c1_wgrad_bn_recompute(x, w, dout, arg, mean, invstd, bn_beta, coef, pool, dw=None, beta=0, nchunk=None):
    """conv3d_c1_wgrad_bn (bf16 MFMA form) with the conv output recomputed per tile from x and the weights w (and the sign of the
    pooled output from it and the BatchNorm bias bn_beta)."""
    _f32(x, w, dout, mean, invstd, bn_beta, coef, dw)
    b, t, h, wd = x.shape
    assert dout.is_contiguous() and w.is_contiguous()
    if nchunk is None:
        cap = int(os.environ.get("MAAVSS_C1_NCHUNK", "1024"))
        nchunk = max(1, min(cap, (b * t * ((h + 15) // 16) * ((wd + 15) // 16)) // 2))
    ws = torch.empty(nchunk * 1200, device=x.device, dtype=torch.float32)
    if dw is None:
        dw, beta = torch.empty(16, 1, 3, 5, 5, device=x.device, dtype=torch.float32), 0
    call("maavss_conv3d_c1_wgrad_bn_recompute", ptr(x), ptr(w), ptr(dout), ptr(arg), ptr(mean), ptr(invstd), ptr(bn_beta), ptr(coef), pool,
         ptr(dw), ptr(ws), nchunk, b, t, h, wd, int(beta), stream_ptr())
    return dw


# ---------------------------------------------------------------------------------------------- batch norm
def bn_stats(y2d_rows, c):
    """y: any contiguous channels-last tensor with last dim c -> partial sums [nblk,2,c]."""
    _f32(y2d_rows)
    rows = y2d_rows.numel() // c
    nblk = query("maavss_bn_stats_nblk", rows)
    part = torch.empty(nblk, 2, c, device=y2d_rows.device, dtype=torch.float32)
    call("maavss_bn_stats", ptr(y2d_rows), ptr(part), rows, c, stream_ptr())
    return part


def bn_finalize(part, count, running_mean=None, running_var=None, num_batches_tracked=None, eps=BN_EPS,
                momentum=BN_MOMENTUM):
    nblk, _, c = part.shape
    mean = torch.empty(c, device=part.device, dtype=torch.float32)
    invstd = torch.empty_like(mean)
    ws = torch.empty(256 * 2 * c, device=part.device, dtype=torch.float32) if nblk > 512 else None
    call("maavss_bn_finalize", ptr(part), nblk, c, float(count), float(eps), float(momentum), ptr(mean), ptr(invstd),
         ptr(running_mean), ptr(running_var), ptr(num_batches_tracked), ptr(ws), stream_ptr())
    return mean, invstd


def bn_finalize_synced(part, count, reduce_fn, running_mean=None, running_var=None, num_batches_tracked=None, eps=BN_EPS,
                        momentum=BN_MOMENTUM):
    """bn_finalize with the per-channel sums (and the element count) all-reduced over the data-parallel ranks by
    `reduce_fn(double tensor [2C+1])` in between: global-batch statistics, as on the reference's single device."""
    nblk, _, c = part.shape
    sums = torch.empty(2 * c + 1, device=part.device, dtype=torch.float64)
    ws = torch.empty(256 * 2 * c, device=part.device, dtype=torch.float32) if nblk > 512 else None
    call("maavss_bn_partials_to_sums", ptr(part), nblk, c, float(count), ptr(sums), ptr(ws), stream_ptr())
    reduce_fn(sums)
    mean = torch.empty(c, device=part.device, dtype=torch.float32)
    invstd = torch.empty_like(mean)
    call("maavss_bn_finalize_sums", ptr(sums), c, float(eps), float(momentum), ptr(mean), ptr(invstd), ptr(running_mean),
         ptr(running_var), ptr(num_batches_tracked), stream_ptr())
    return mean, invstd


def bn_eval_stats(running_mean, running_var, eps=BN_EPS):
    """eval-mode BatchNorm statistics (running mean / var) in the (mean, invstd) form the fused kernels take."""
    _f32(running_mean, running_var)
    mean, invstd = torch.empty_like(running_mean), torch.empty_like(running_mean)
    call("maavss_bn_eval_stats", ptr(running_mean), ptr(running_var), float(eps), ptr(mean), ptr(invstd),
         running_mean.numel(), stream_ptr())
    return mean, invstd


def cl_strides(t, hp, wp, c):
    """element strides (b, t, pos, c) of a channels-last pooled tensor [B,T,Hp,Wp,C]."""
    return (t * hp * wp * c, hp * wp * c, c, 1)


def bn_pool_act_fwd(y, mean, invstd, gamma, beta, pool, act, out=None, strides=None, want16=False, want_bf16=False):
    """y [B,T,H,W,C] -> out (default channels-last [B,T,H//p,W//p,C]) and argmax (uint8) when pool > 1.
    want16: also return an IEEE-half copy of the pooled activation (the next Conv3d's forward MFMA operand); want_bf16: and / or a
    bfloat16 copy (the next Conv3d's weight-gradient operand, conv3d_wgrad) -- appended to the returned tuple in that order."""
    _f32(y, mean, invstd, gamma, beta)
    b, t, h, w, c = y.shape
    hp, wp = h // pool, w // pool
    if out is None:
        out = torch.empty(b, t, hp, wp, c, device=y.device, dtype=torch.float32)
        strides = cl_strides(t, hp, wp, c)
    arg = torch.empty(b, t, hp, wp, c, device=y.device, dtype=torch.uint8) if pool > 1 else None
    out16 = torch.empty(b, t, hp, wp, c, device=y.device, dtype=torch.float16) if want16 else None
    outb = torch.empty(b, t, hp, wp, c, device=y.device, dtype=torch.bfloat16) if want_bf16 else None
    call("maavss_bn_pool_act_fwd", ptr(y), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), ptr(out), ptr(arg), b, t, h,
         w, c, pool, act, *[int(s) for s in strides], ptr(out16), ptr(outb), stream_ptr())
    return (out, arg) + ((out16,) if want16 else ()) + ((outb,) if want_bf16 else ())


def bn_pool_act_bwd(dout, out, arg, y, mean, invstd, gamma, pool, act, strides=None, dgamma=None, dbeta=None,
                    accumulate=False, dy=None, coef_only=False, beta=None, reduce_fn=None, dy_bf16=False):
    """`reduce_fn` (global-batch BatchNorm under data parallelism): all-reduces the [2C+1] double sums between the
    reduction and the dx pass; dgamma / dbeta keep this rank's sums (the gradient all-reduce adds the ranks)."""
    _f32(y, mean, invstd, gamma)
    b, t, h, w, c = y.shape
    hp, wp = h // pool, w // pool
    if strides is None:
        strides = cl_strides(t, hp, wp, c)
    nblk = query("maavss_bn_stats_nblk", b * t * hp * wp)
    ws = torch.empty(2 * c * nblk + 3 * c, device=y.device, dtype=torch.float32)
    if dy is None and not coef_only:
        dy = torch.empty(y.shape, device=y.device, dtype=torch.bfloat16 if dy_bf16 else torch.float32)
    _f32(beta)
    if reduce_fn is not None:
        geom = (b, t, h, w, c, pool, act, *[int(s) for s in strides], stream_ptr())
        local = torch.empty(2 * c + 1, device=y.device, dtype=torch.float64)
        call("maavss_bn_pool_act_bwd_sums", ptr(dout), ptr(out), ptr(arg), ptr(y), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta),
             ptr(ws), ptr(local), *geom)
        glob = local.clone()
        reduce_fn(glob)
        coef = ws[2 * c * nblk:]
        call("maavss_bn_pool_act_bwd_finish", ptr(dout), ptr(out), ptr(arg), ptr(y), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta),
             ptr(dy), ptr(dgamma), ptr(dbeta), int(accumulate), ptr(local), ptr(glob), ptr(coef), *geom[:-1], int(dy_bf16), geom[-1])
        return coef.view(3, c) if coef_only else dy
    call("maavss_bn_pool_act_bwd", ptr(dout), ptr(out), ptr(arg), ptr(y), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), ptr(dy),
         ptr(dgamma), ptr(dbeta), int(accumulate), ptr(ws), b, t, h, w, c, pool, act, *[int(s) for s in strides],
         int(dy_bf16), stream_ptr())
    if coef_only:      # dgamma / dbeta done; the consumer folds dx in (conv3d_c1_wgrad_bn): [3, C] coefficients
        return ws[2 * c * nblk:].view(3, c)
    return dy


# ---------------------------------------------------------------------------------------------- conv2d
def conv2d_out(h, w, sh, sw, pw):
    return (h + 2 - 3) // sh + 1, (w + 2 * pw - 9) // sw + 1


def conv2d_fwd(x, w, stride, pw, in_nchw):
    _f32(x, w)
    co, ci = w.shape[0], w.shape[1]
    if in_nchw:
        b, _, h, wd = x.shape
    else:
        b, h, wd, _ = x.shape
    ho, wo = conv2d_out(h, wd, stride[0], stride[1], pw)
    y = torch.empty(b, ho, wo, co, device=x.device, dtype=torch.float32)
    call("maavss_conv2d_fwd", ptr(x), ptr(w), ptr(y), b, ci, h, wd, co, stride[0], stride[1], pw, 0 if in_nchw else 1,
         stream_ptr())
    return y


def conv2d_dgrad(dy, w, in_hw, stride, pw):
    _f32(dy, w)
    co, ci = w.shape[0], w.shape[1]
    b = dy.shape[0]
    dx = torch.empty(b, in_hw[0], in_hw[1], ci, device=dy.device, dtype=torch.float32)
    call("maavss_conv2d_dgrad", ptr(dy), ptr(w), ptr(dx), b, ci, in_hw[0], in_hw[1], co, stride[0], stride[1], pw,
         stream_ptr())
    return dx


def conv2d_wgrad(x, dy, w_shape, stride, pw, in_nchw, dw=None, beta=0):
    _f32(x, dy, dw)
    co, ci = w_shape[0], w_shape[1]
    if in_nchw:
        b, _, h, wd = x.shape
    else:
        b, h, wd, _ = x.shape
    ho, wo = dy.shape[1], dy.shape[2]
    nchunk = query("maavss_conv2d_wgrad_nchunk", b, ho, wo, ci, co)
    ws = torch.empty(nchunk * co * ci * 27, device=x.device, dtype=torch.float32)
    if dw is None:
        dw = torch.empty(co, ci, 3, 9, device=x.device, dtype=torch.float32)
        beta = 0
    call("maavss_conv2d_wgrad", ptr(x), ptr(dy), ptr(dw), ptr(ws), b, ci, h, wd, co, stride[0], stride[1], pw,
         0 if in_nchw else 1, int(beta), stream_ptr())
    return dw


# ---------------------------------------------------------------------------------------------- convt2d (STFT decoder)
def convt2d_out(h, w, kw, stride, opad):
    return (h - 1) * stride[0] + 1 + opad[0], (w - 1) * stride[1] - 8 + kw + opad[1]


def convt2d_fwd(x, w, stride, opad, out_nhwc):
    """x NHWC [B,Hi,Wi,Ci], w [Ci,Co,3,kw] -> y NHWC [B,Ho,Wo,Co] or NCHW [B,Co,Ho,Wo] (last decoder layer)."""
    _f32(x, w)
    b, hi, wi, ci = x.shape
    assert w.shape[0] == ci and w.shape[2] == 3
    co, kw = w.shape[1], w.shape[3]
    ho, wo = convt2d_out(hi, wi, kw, stride, opad)
    y = torch.empty((b, ho, wo, co) if out_nhwc else (b, co, ho, wo), device=x.device, dtype=torch.float32)
    call("maavss_convt2d_fwd", ptr(x), ptr(w), ptr(y), b, ci, hi, wi, co, kw, stride[0], stride[1], opad[0], opad[1],
         1 if out_nhwc else 0, stream_ptr())
    return y


def convt2d_dgrad(dy, w, in_hw, stride, opad, out_nhwc):
    _f32(dy, w)
    ci, co, kw = w.shape[0], w.shape[1], w.shape[3]
    b = dy.shape[0]
    dx = torch.empty(b, in_hw[0], in_hw[1], ci, device=dy.device, dtype=torch.float32)
    call("maavss_convt2d_dgrad", ptr(dy), ptr(w), ptr(dx), b, ci, in_hw[0], in_hw[1], co, kw, stride[0], stride[1], opad[0],
         opad[1], 1 if out_nhwc else 0, stream_ptr())
    return dx


def convt2d_wgrad(x, dy, w_shape, stride, opad, out_nhwc):
    _f32(x, dy)
    b, hi, wi, ci = x.shape
    co, kw = w_shape[1], w_shape[3]
    nchunk = query("maavss_convt2d_wgrad_nchunk", b, hi, wi)
    ws = torch.empty(nchunk * ci * co * 3 * kw, device=x.device, dtype=torch.float32)
    dw = torch.empty(ci, co, 3, kw, device=x.device, dtype=torch.float32)
    call("maavss_convt2d_wgrad", ptr(x), ptr(dy), ptr(dw), ptr(ws), b, ci, hi, wi, co, kw, stride[0], stride[1], opad[0],
         opad[1], 1 if out_nhwc else 0, 0, stream_ptr())
    return dw


# ---------------------------------------------------------------------------------------------- generic conv2d (K19)
import ctypes as _ct

ACT_LEAKY = 3


class Map:
    """A [B, H, W, C] view of a tensor for the generic convolution kernels: `t` is stored NHWC (possibly with more
    channels allocated than `c`, for the BatchNorm kernels) or NCHW (the network's own tensors)."""

    def __init__(self, t, nchw=False, c=None):
        _f32(t)
        self.t, self.nchw = t, nchw
        if nchw:
            self.b, self.c_alloc, self.h, self.w = t.shape
            st = (t.stride(0), t.stride(2), t.stride(3), t.stride(1))
        else:
            self.b, self.h, self.w, self.c_alloc = t.shape
            st = (t.stride(0), t.stride(1), t.stride(2), t.stride(3))
        self.c = self.c_alloc if c is None else c
        self.strides = (_ct.c_int64 * 4)(*st)


def _cgen_args(small, big, w, stride, pad):
    cs, cb, kh, kw = w.shape
    assert small.c == cs and big.c == cb and small.b == big.b, (small.c, cs, big.c, cb)
    return (small.b, cs, small.h, small.w, cb, big.h, big.w, kh, kw, stride[0], stride[1], pad[0], pad[1], small.strides, big.strides)


def conv_gen_small(big, w, bias, small, stride, pad):
    """small = bias + conv(big, w): Conv2d forward (w [Co,Ci,kh,kw]) / ConvTranspose2d input gradient (bias None)."""
    _f32(w, bias)
    call("maavss_conv2d_gen_small", ptr(big.t), ptr(w), ptr(bias), ptr(small.t), *_cgen_args(small, big, w, stride, pad), stream_ptr())


def conv_gen_big(small, w, bias, big, stride, pad):
    """big = bias + convT(small, w): ConvTranspose2d forward (w [Ci,Co,kh,kw]) / Conv2d input gradient (bias None)."""
    _f32(w, bias)
    call("maavss_conv2d_gen_big", ptr(small.t), ptr(w), ptr(bias), ptr(big.t), *_cgen_args(small, big, w, stride, pad), stream_ptr())


def conv_gen_wgrad(small, big, w_shape, stride, pad, dw=None, beta=0):
    cs, cb, kh, kw = w_shape
    nchunk = query("maavss_conv2d_gen_wgrad_nchunk", small.b, small.h, small.w)
    ws = torch.empty(nchunk * cs * cb * kh * kw, device=small.t.device, dtype=torch.float32)
    if dw is None:
        dw, beta = torch.empty(cs, cb, kh, kw, device=small.t.device, dtype=torch.float32), 0
    assert small.c == cs and big.c == cb
    call("maavss_conv2d_gen_wgrad", ptr(small.t), ptr(big.t), ptr(dw), ptr(ws), small.b, cs, small.h, small.w, cb, big.h, big.w, kh, kw,
         stride[0], stride[1], pad[0], pad[1], small.strides, big.strides, int(beta), stream_ptr())
    return dw


def channel_sum(m, out=None, beta=0):
    """sum over batch and positions of a Map -> [C] (bias gradient of a convolution)."""
    if out is None:
        out, beta = torch.empty(m.c, device=m.t.device, dtype=torch.float32), 0
    if m.nchw:      # rows are not uniformly strided across (b, y*x): reduce per batch image
        for bi in range(m.b):
            call("maavss_channel_sum", ptr(m.t[bi]), ptr(out), m.h * m.w, m.c, 1, m.h * m.w, 1 if (beta or bi) else 0, stream_ptr())
    else:
        assert m.t.is_contiguous()
        call("maavss_channel_sum", ptr(m.t), ptr(out), m.b * m.h * m.w, m.c, m.c_alloc, 1, int(beta), stream_ptr())
    return out


def rows_sum(x2d, out=None, beta=0):
    """column sums of a contiguous [rows, n] matrix (bias gradient of a Linear layer)."""
    _f32(x2d)
    rows, n = x2d.shape
    if out is None:
        out, beta = torch.empty(n, device=x2d.device, dtype=torch.float32), 0
    call("maavss_channel_sum", ptr(x2d), ptr(out), rows, n, n, 1, int(beta), stream_ptr())
    return out


def bias_act_(z, bias, act=ACT_NONE, slope=0.3):
    _f32(z, bias)
    rows, n = z.shape
    call("maavss_bias_act_fwd", ptr(z), ptr(bias), rows, n, act, float(slope), stream_ptr())
    return z


def leaky_bwd(dout, out, slope=0.3):
    _f32(dout, out)
    dz = torch.empty_like(out)
    call("maavss_leaky_bwd", ptr(dout), ptr(out), ptr(dz), out.numel(), float(slope), stream_ptr())
    return dz


# ---------------------------------------------------------------------------------------------- lstm
def lstm_fwd(gx, whh_f, whh_b):
    """gx [B,L,2,4,256] -> av [B,L,512] and the saved state (hp, gs, cs)."""
    _f32(gx, whh_f, whh_b)
    b, l = gx.shape[0], gx.shape[1]
    dev = gx.device
    av = torch.empty(b, l, 512, device=dev, dtype=torch.float32)
    hp = torch.empty(b, l, 2, 256, device=dev, dtype=torch.float32)
    gs = torch.empty(b, l, 2, 4, 256, device=dev, dtype=torch.float32)
    cs = torch.empty(b, l, 2, 256, device=dev, dtype=torch.float32)
    call("maavss_lstm_fwd", ptr(gx), ptr(whh_f), ptr(whh_b), ptr(av), ptr(hp), ptr(gs), ptr(cs), b, l, stream_ptr())
    return av, hp, gs, cs


def lstm_bwd(dav, whh_f, whh_b, gs, cs):
    _f32(dav, whh_f, whh_b, gs, cs)
    b, l = dav.shape[0], dav.shape[1]
    dgx = torch.empty(b, l, 2, 4, 256, device=dav.device, dtype=torch.float32)
    dc = torch.empty(2, b, 256, device=dav.device, dtype=torch.float32)
    call("maavss_lstm_bwd", ptr(dav), ptr(whh_f), ptr(whh_b), ptr(gs), ptr(cs), ptr(dgx), ptr(dc), b, l, stream_ptr())
    return dgx


# ---------------------------------------------------------------------------------------------- loss / optimiser
def act_bwd(dout, out, act):
    _f32(dout, out)
    dz = torch.empty_like(out)
    call("maavss_act_bwd", ptr(dout), ptr(out), ptr(dz), out.numel(), act, stream_ptr())
    return dz


def mse_pair(a_pred, a_tgt, v_pred, v_tgt, coeff, num_seq=1, want_grads=True):
    """-> losses [3] (a_loss, v_loss, total) and d(total)/d(a_pred), d(total)/d(v_pred)."""
    _f32(a_pred, a_tgt, v_pred, v_tgt)
    assert a_pred.shape == a_tgt.shape and v_pred.shape == v_tgt.shape
    dev = a_pred.device
    d_a = torch.empty_like(a_pred) if want_grads else None
    d_v = torch.empty_like(v_pred) if want_grads else None
    losses = torch.empty(3, device=dev, dtype=torch.float32)
    ws = torch.empty(1024, device=dev, dtype=torch.float32)
    call("maavss_mse_pair", ptr(a_pred), ptr(a_tgt), a_pred.numel(), ptr(v_pred), ptr(v_tgt), v_pred.numel(),
         float(coeff), 1.0 / num_seq, ptr(d_a), ptr(d_v), ptr(losses), ptr(ws), stream_ptr())
    return losses, d_a, d_v


def adam_step(p, g, m, v, lr, step, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
    _f32(p, g, m, v)
    call("maavss_adam_step", ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), float(lr), float(betas[0]), float(betas[1]),
         float(eps), int(step), float(grad_scale), stream_ptr())


def f32_to_bf16(src, dst):
    """bf16 wire format of the gradient all-reduce: dst (bf16) = round(src (f32)); length a multiple of 8."""
    _lib.require_cuda(src, dst)
    assert src.dtype == torch.float32 and dst.dtype == torch.bfloat16 and src.numel() == dst.numel()
    call("maavss_f32_to_bf16", ptr(src), ptr(dst), src.numel(), stream_ptr())


def bf16_to_f32(src, dst):
    _lib.require_cuda(src, dst)
    assert src.dtype == torch.bfloat16 and dst.dtype == torch.float32 and src.numel() == dst.numel()
    call("maavss_bf16_to_f32", ptr(src), ptr(dst), src.numel(), stream_ptr())
