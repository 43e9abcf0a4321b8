"""GPU unit parity of every HIP kernel family (through the C-ABI) against plain torch fp32 on the CPU.

precise=True  : exact-f32 MFMA / VALU -> fp32 tolerances.
precise=False : operands rounded to bf16 inside the kernel; the reference is computed in fp32 from the SAME
                bf16-rounded operands, so only accumulation order differs and the tolerance stays tight.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rt(x, mode=False):
    """round to bf16 (mode 0/False) or IEEE half (mode 2) and back: what the 16-bit MFMA modes do to operands"""
    return x.to(torch.float16 if mode == 2 else torch.bfloat16).to(torch.float32)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(got, want, rtol, atol, msg=""):
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=rtol, atol=atol, err_msg=msg)


# ----------------------------------------------------------------------------------------------- gemm
@pytest.mark.parametrize("precise", [True, False])
@pytest.mark.parametrize("m,n,k,ta,tb", [
    (2, 4096, 8192, False, False),     # fc1 forward at B=2 (skinny, split-K)
    (32, 512, 4096, False, False),     # fc2
    (32, 4112, 512, False, False),     # a_fc1 (N not a multiple of the tile)
    (512, 2048, 288, False, False),    # LSTM input projection, 224^2 variant (K % 32 != 0)
    (8192, 4096, 2, True, True),       # dW = dY^T X at B=2 (K=2)
    (32, 8192, 4096, False, True),     # dX = dY W
    (1024, 256, 512, True, True),      # dW_hh
    (130, 70, 100, False, False),      # ragged everything
    (32, 4096, 8192, False, False),    # fc1 forward at the benched batch: weight-streaming form, 16 K slices (atomics)
    (32, 12544, 512, False, False),    # v_fc1 forward (112^2): one K slice, direct stores
    (32, 512, 12544, False, True),     # v_fc1 input gradient: 98 N slices
    (4096, 8192, 32, True, True),      # fc1 weight gradient at the benched batch
    (12544, 512, 32, True, True),      # v_fc1 weight gradient
    (5, 1000, 528, False, False),      # skinny forms, ragged: rows, columns and the last K slice
    (7, 780, 300, False, True),
    (300, 780, 7, True, True),
])
def test_gemm(m, n, k, ta, tb, precise):
    from maavss_amd import ops
    a = rnd(*((k, m) if ta else (m, k)), seed=1)
    b = rnd(*((k, n) if tb else (n, k)), seed=2)
    ar, br = (a, b) if precise else (rt(a), rt(b))
    want = (ar.t() if ta else ar).double() @ (br if tb else br.t()).double()
    got = ops.gemm(a.cuda(), b.cuda(), ta, tb, precise=precise)
    tol = 2e-5 * np.sqrt(k) + 1e-5
    close(got, want.float(), 1e-4, tol)


def test_gemm_epilogues():
    from maavss_amd import ops
    a, b = rnd(32, 512, seed=3, scale=0.2), rnd(300, 512, seed=4, scale=0.2)
    z = a @ b.t()
    close(ops.gemm(a.cuda(), b.cuda(), act=ops.ACT_TANH, precise=True), torch.tanh(z), 1e-4, 1e-5)
    close(ops.gemm(a.cuda(), b.cuda(), act=ops.ACT_SIGMOID, precise=True, split_k=4), torch.sigmoid(z), 1e-4, 1e-5)
    c0 = rnd(32, 300, seed=5)
    out = c0.clone().cuda()
    ops.gemm(a.cuda(), b.cuda(), out=out, beta=1, precise=True)
    close(out, c0 + z, 1e-4, 1e-5)
    outt = ops.gemm(a.cuda(), b.cuda(), precise=True, trans_c=True)
    close(outt, z.t().contiguous(), 1e-4, 1e-5)
    # the weight-streaming forms of the Linear layers (linear_skinny.hip): activation and accumulate epilogues, with and without
    # a K / N split (= atomics + separate activation pass)
    for n, k in ((1024, 512), (1024, 2048)):
        a, b = rnd(32, k, seed=6, scale=0.2), rnd(n, k, seed=7, scale=0.1)
        z = a @ b.t()
        close(ops.gemm(a.cuda(), b.cuda(), act=ops.ACT_SIGMOID, precise=True), torch.sigmoid(z), 1e-4, 2e-5)
        c0 = rnd(32, n, seed=8)
        out = c0.clone().cuda()
        ops.gemm(a.cuda(), b.cuda(), out=out, beta=1, precise=True)
        close(out, c0 + z, 1e-4, 2e-5)
        dy = rnd(32, n, seed=9, scale=0.2)
        dx0 = rnd(32, k, seed=10)
        out = dx0.clone().cuda()
        ops.gemm(dy.cuda(), b.cuda(), trans_b=True, out=out, beta=1, precise=True)          # dX += dY W
        close(out, dx0 + dy @ b, 1e-4, 5e-5)
        dw0 = rnd(n, k, seed=11)
        out = dw0.clone().cuda()
        ops.gemm(dy.cuda(), a.cuda(), trans_a=True, trans_b=True, out=out, beta=1, precise=True)   # dW += dY^T X
        close(out, dw0 + dy.t() @ a, 1e-4, 2e-5)


# ----------------------------------------------------------------------------------------------- conv3d
def to_cl(x):      # NCDHW -> channels-last [B,T,H,W,C]
    return x.permute(0, 2, 3, 4, 1).contiguous()


def from_cl(x):
    return x.permute(0, 4, 1, 2, 3).contiguous()


@pytest.mark.parametrize("precise", [1, 0, 2])
@pytest.mark.parametrize("ci,co,pad,b,t,h,w", [
    (16, 32, 2, 1, 3, 20, 36),
    (32, 64, 2, 2, 2, 16, 16),
    (64, 64, 2, 1, 4, 28, 28),
    (64, 16, 3, 2, 3, 10, 10),
    (64, 64, 2, 2, 1, 16, 16),     # T = 1: only the centre kd plane touches a real frame (prefetch range kd_lo = kd_hi)
])
def test_conv3d_igemm_fwd_dgrad_wgrad(ci, co, pad, b, t, h, w, precise):
    from maavss_amd import ops
    x = rnd(b, ci, t, h, w, seed=1)
    wgt = rnd(co, ci, 3, 5, 5, seed=2, scale=(ci * 75) ** -0.5)
    xr, wr = (x, wgt) if precise == 1 else (rt(x, precise), rt(wgt, precise))
    xr = xr.clone().requires_grad_(True)
    wr = wr.clone().requires_grad_(True)
    y_ref = F.conv3d(xr, wr, padding=(1, pad, pad))
    dy = rnd(*y_ref.shape, seed=3)
    dyr = dy if precise == 1 else rt(dy, precise)
    tol = dict(rtol=2e-4, atol=2e-4)

    x_cl = to_cl(x).cuda()
    wt = ops.conv3d_prep(wgt.cuda(), 0, precise)
    y, part = ops.conv3d_igemm(x_cl, wt, co, pad, precise, want_stats=True)
    close(from_cl(y), y_ref, **tol)
    # fused BatchNorm partial sums
    s = part.sum(0).cpu()
    yr = y_ref.detach()
    np.testing.assert_allclose(s[0].numpy(), yr.sum((0, 2, 3, 4)).numpy(), rtol=1e-3, atol=1e-2)
    np.testing.assert_allclose(s[1].numpy(), (yr * yr).sum((0, 2, 3, 4)).numpy(), rtol=1e-3, atol=1e-2)

    # input gradient = the same kernel on flipped / transposed weights with pad 4-p
    gx_ref, = torch.autograd.grad(y_ref, xr, dyr, retain_graph=True)
    wtd = ops.conv3d_prep(wgt.cuda(), 1, precise)
    dx, _ = ops.conv3d_igemm(to_cl(dy).cuda(), wtd, ci, 4 - pad, precise)
    close(from_cl(dx), gx_ref, **tol)

    gw_ref, = torch.autograd.grad(y_ref, wr, dyr)
    dw = ops.conv3d_wgrad(x_cl, to_cl(dy).cuda(), pad, precise)
    scale = gw_ref.abs().max().item()
    close(dw, gw_ref, 2e-4, 2e-4 * scale + 1e-5)
    dw2 = ops.conv3d_wgrad(x_cl, to_cl(dy).cuda(), pad, precise, dw=dw.clone(), beta=1, nchunk=3)
    close(dw2, 2 * gw_ref, 2e-4, 4e-4 * scale + 1e-5)


@pytest.mark.parametrize("ci,co,pad,b,t,h,w", [
    (16, 32, 2, 1, 3, 20, 36),     # C_in = 16 / 32: LDS-DMA halo (zero source = the padded tail of the weight rows)
    (32, 64, 2, 2, 2, 16, 16),
    (64, 64, 2, 1, 4, 28, 28),     # C_in = 64: register prefetch of 16-bit vectors
    (64, 16, 3, 2, 3, 10, 10),
    (32, 16, 2, 2, 2, 23, 17),     # input-gradient shape of the first MFMA layer, ragged tiles
])
def test_conv3d_16bit_operand_storage_is_bit_identical(ci, co, pad, b, t, h, w):
    """Operands that arrive already in the MFMA format (x as IEEE half / bf16, dy as bf16: rounded once by the producer
    kernels bn_pool_act_fwd / bn_pool_act_bwd) must give exactly the result of the f32-input path, which applies the same
    rounding while staging."""
    from maavss_amd import ops
    x = to_cl(rnd(b, ci, t, h, w, seed=1)).cuda()
    wgt = rnd(co, ci, 3, 5, 5, seed=2, scale=(ci * 75) ** -0.5).cuda()
    for mode, tdt in ((ops.MODE_F16, torch.float16), (ops.MODE_BF16, torch.bfloat16)):
        wt = ops.conv3d_prep(wgt, 0, mode)
        y32, p32 = ops.conv3d_igemm(x, wt, co, pad, mode, want_stats=True)
        y16, p16 = ops.conv3d_igemm(x.to(tdt), wt, co, pad, mode, want_stats=True)
        assert torch.equal(y32, y16) and torch.equal(p32, p16), (ci, co, mode)
    ho, wo = h + 2 * pad - 4, w + 2 * pad - 4
    dy = rnd(b, t, ho, wo, co, seed=3).cuda()
    if (ci, co) in ((16, 32), (32, 64), (64, 64), (64, 16)):                          # the model's weight-gradient shapes
        dw32 = ops.conv3d_wgrad(x, dy, pad, ops.MODE_BF16)
        dw16 = ops.conv3d_wgrad(x, dy.to(torch.bfloat16), pad, ops.MODE_BF16)
        assert torch.equal(dw32, dw16)
        if (ci, co) in ops.WGRAD_X16_SHAPES:      # x as the bf16 copy of its producer: both images by LDS-DMA, two image pairs
            dwx = ops.conv3d_wgrad(x.to(torch.bfloat16), dy.to(torch.bfloat16), pad, ops.MODE_BF16)
            assert torch.equal(dw32, dwx)
            dwx3 = ops.conv3d_wgrad(x.to(torch.bfloat16), dy.to(torch.bfloat16), pad, ops.MODE_BF16, nchunk=3)      # several tiles per chunk: both pairs used
            torch.testing.assert_close(dwx3, dw32, rtol=1e-5, atol=1e-6 * dw32.abs().max().item())
    if (co, ci) in ((32, 16), (64, 32), (64, 64), (16, 64), (16, 32), (32, 64)):      # dgrad instantiations
        wtd = ops.conv3d_prep(wgt, 1, ops.MODE_BF16)
        dx32, _ = ops.conv3d_igemm(dy, wtd, ci, 4 - pad, ops.MODE_BF16)
        dx16, _ = ops.conv3d_igemm(dy.to(torch.bfloat16), wtd, ci, 4 - pad, ops.MODE_BF16)
        assert torch.equal(dx32, dx16)


def test_bn_pool_act_16bit_side_outputs():
    """bn_pool_act_fwd's IEEE-half copy == the f32 output rounded; bn_pool_act_bwd(dy_bf16) == its f32 dy rounded"""
    from maavss_amd import ops
    b, t, c, h, w, pool = 2, 3, 32, 12, 20, 2
    y = rnd(b, t, h, w, c, seed=1).cuda()
    part = ops.bn_stats(y, c)
    mean, invstd = ops.bn_finalize(part, b * t * h * w)
    gamma, beta = (1 + 0.3 * rnd(c, seed=2)).cuda(), (0.2 * rnd(c, seed=3)).cuda()
    out, arg, out16, outb = ops.bn_pool_act_fwd(y, mean, invstd, gamma, beta, pool, 0, want16=True, want_bf16=True)
    out_b, arg_b = ops.bn_pool_act_fwd(y, mean, invstd, gamma, beta, pool, 0)
    assert torch.equal(out, out_b) and torch.equal(arg, arg_b)
    assert out16.dtype == torch.float16 and torch.equal(out16, out.half())
    assert outb.dtype == torch.bfloat16 and torch.equal(outb, out.bfloat16())
    _, _, only_b = ops.bn_pool_act_fwd(y, mean, invstd, gamma, beta, pool, 0, want_bf16=True)
    assert torch.equal(only_b, outb)
    dout = rnd(*out.shape, seed=4).cuda()
    dg, db = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
    dy32 = ops.bn_pool_act_bwd(dout, out, arg, y, mean, invstd, gamma, pool, 0, dgamma=dg, dbeta=db, beta=beta)
    dg2, db2 = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
    dy16 = ops.bn_pool_act_bwd(dout, out, arg, y, mean, invstd, gamma, pool, 0, dgamma=dg2, dbeta=db2, beta=beta, dy_bf16=True)
    assert dy16.dtype == torch.bfloat16 and torch.equal(dy16, dy32.bfloat16())
    assert torch.equal(dg, dg2) and torch.equal(db, db2)


def test_conv3d_c1():
    from maavss_amd import ops
    b, t, h, w = 2, 3, 40, 24
    x = rnd(b, 1, t, h, w, seed=1).requires_grad_(True)
    wgt = rnd(16, 1, 3, 5, 5, seed=2, scale=0.1).requires_grad_(True)
    y_ref = F.conv3d(x, wgt, padding=(1, 2, 2))
    dy = rnd(*y_ref.shape, seed=3)
    gw_ref, = torch.autograd.grad(y_ref, wgt, dy)
    xc = x.detach()[:, 0].contiguous().cuda()
    y, part = ops.conv3d_c1_fwd(xc, wgt.detach().cuda(), want_stats=True)
    close(from_cl(y), y_ref, 1e-5, 1e-5)
    s = part.sum(0).cpu()
    np.testing.assert_allclose(s[0].numpy(), y_ref.detach().sum((0, 2, 3, 4)).numpy(), rtol=1e-3, atol=1e-2)
    dw = ops.conv3d_c1_wgrad(xc, to_cl(dy).cuda())
    close(dw, gw_ref, 1e-4, 1e-4 * gw_ref.abs().max().item())
    # the MFMA form of the forward pass (IEEE-half operands, f32 accumulation) against torch on the same rounded operands
    y16, part16 = ops.conv3d_c1_fwd(xc, wgt.detach().cuda(), want_stats=True, precise=ops.MODE_F16)
    y16_ref = F.conv3d(x.detach().half().float(), wgt.detach().half().float(), padding=(1, 2, 2))
    close(from_cl(y16), y16_ref, 2e-5, 2e-5)
    s16 = part16.sum(0).cpu()
    np.testing.assert_allclose(s16[0].numpy(), y16_ref.sum((0, 2, 3, 4)).numpy(), rtol=1e-3, atol=1e-2)
    np.testing.assert_allclose(s16[1].numpy(), (y16_ref * y16_ref).sum((0, 2, 3, 4)).numpy(), rtol=1e-3, atol=1e-2)


# ----------------------------------------------------------------------------------------------- BN + pool + act
@pytest.mark.parametrize("c,pool,act,h,w", [(16, 2, 0, 12, 20), (64, 3, 0, 28, 28), (16, 3, 0, 11, 11), (8, 1, 1, 9, 17)])
def test_bn_pool_act(c, pool, act, h, w):
    from maavss_amd import ops
    b, t = 2, 3
    y = rnd(b, c, t, h, w, seed=1).requires_grad_(True)
    gamma = (1 + 0.3 * rnd(c, seed=2)).requires_grad_(True)
    beta = (0.2 * rnd(c, seed=3)).requires_grad_(True)
    rm, rv = 0.1 * rnd(c, seed=4), 1 + 0.1 * rnd(c, seed=5).abs()
    rm_ref, rv_ref = rm.clone(), rv.clone()
    z = F.batch_norm(y, rm_ref, rv_ref, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    if pool > 1:
        z = F.max_pool3d(z, (1, pool, pool))
    out_ref = torch.tanh(z) if act == 1 else F.leaky_relu(z, 0.01)
    dout = rnd(*out_ref.shape, seed=6)
    gy, gg, gb = torch.autograd.grad(out_ref, (y, gamma, beta), dout)

    y_cl = to_cl(y.detach()).cuda()
    part = ops.bn_stats(y_cl, c)
    rmc, rvc = rm.cuda(), rv.cuda()
    nbt = torch.zeros((), dtype=torch.long, device="cuda")
    mean, invstd = ops.bn_finalize(part, b * t * h * w, rmc, rvc, nbt)
    close(rmc, rm_ref, 1e-5, 1e-6)
    close(rvc, rv_ref, 1e-5, 1e-6)
    assert nbt.item() == 1
    out, arg = ops.bn_pool_act_fwd(y_cl, mean, invstd, gamma.detach().cuda(), beta.detach().cuda(), pool, act)
    close(from_cl(out), out_ref, 1e-5, 1e-5)
    dgamma = torch.zeros(c, device="cuda")
    dbeta = torch.zeros(c, device="cuda")
    dy = ops.bn_pool_act_bwd(to_cl(dout).cuda(), out, arg, y_cl, mean, invstd, gamma.detach().cuda(), pool, act,
                             dgamma=dgamma, dbeta=dbeta)
    close(from_cl(dy), gy, 1e-4, 2e-5)
    close(dgamma, gg, 1e-4, 1e-4)
    close(dbeta, gb, 1e-4, 1e-4)
    # with beta: LeakyReLU layers recover xhat at the pooled maximum from `out` instead of gathering y (same numbers);
    # one channel with gamma ~ 0 has to keep the gather
    gam2 = gamma.detach().clone()
    gam2[1] = 1e-4
    for gam in (gamma.detach(), gam2):
        z2 = F.batch_norm(y.detach(), None, None, gam, beta.detach(), training=True, eps=1e-5).requires_grad_(False)
        yq, gq, bq = y.detach().clone().requires_grad_(True), gam.clone().requires_grad_(True), beta.detach().clone().requires_grad_(True)
        zz = F.batch_norm(yq, None, None, gq, bq, training=True, eps=1e-5)
        if pool > 1:
            zz = F.max_pool3d(zz, (1, pool, pool))
        oo = torch.tanh(zz) if act == 1 else F.leaky_relu(zz, 0.01)
        gy2, gg2, gb2 = torch.autograd.grad(oo, (yq, gq, bq), dout)
        out2, arg2 = ops.bn_pool_act_fwd(y_cl, mean, invstd, gam.cuda(), beta.detach().cuda(), pool, act)
        dg2, db2 = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda")
        dy2 = ops.bn_pool_act_bwd(to_cl(dout).cuda(), out2, arg2, y_cl, mean, invstd, gam.cuda(), pool, act, dgamma=dg2, dbeta=db2,
                                  beta=beta.detach().cuda())
        close(from_cl(dy2), gy2, 1e-4, 2e-5)
        close(dg2, gg2, 1e-4, 1e-4)
        close(db2, gb2, 1e-4, 1e-4)


@pytest.mark.parametrize("b,t,h,w", [(2, 3, 40, 24), (1, 2, 33, 50)])
def test_conv3d_c1_wgrad_bn_mfma_matches_the_f32_kernel(b, t, h, w):
    """First-layer weight gradient with the BatchNorm / pool / LeakyReLU backward fused: the MFMA form (x and dy rounded to
    bf16, positions as the K dimension) against the exact-f32 VALU kernel on the same inputs -- they differ by the operand
    rounding only (2^-9 relative per element, averaged over b*t*h*w positions)."""
    from maavss_amd import ops
    x = torch.rand(b, t, h, w, generator=torch.Generator().manual_seed(1)).cuda()
    wgt = rnd(16, 1, 3, 5, 5, seed=2, scale=0.1).cuda()
    y, part = ops.conv3d_c1_fwd(x, wgt, want_stats=True)
    mean, invstd = ops.bn_finalize(part, b * t * h * w)
    gamma, beta = (1 + 0.3 * rnd(16, seed=3)).cuda(), (0.2 * rnd(16, seed=4)).cuda()
    pool = 2
    out, arg = ops.bn_pool_act_fwd(y, mean, invstd, gamma, beta, pool, ops.BN_LEAKY)
    dout = rnd(*out.shape, seed=5).cuda()
    dg, db = torch.zeros(16, device="cuda"), torch.zeros(16, device="cuda")
    coef = ops.bn_pool_act_bwd(dout, out, arg, y, mean, invstd, gamma, pool, ops.BN_LEAKY, dgamma=dg, dbeta=db, beta=beta, coef_only=True)
    dw32 = ops.conv3d_c1_wgrad_bn(x, y, dout, out, arg, mean, invstd, coef, pool, nchunk=5)
    dw16 = ops.conv3d_c1_wgrad_bn(x, y, dout, out, arg, mean, invstd, coef, pool, nchunk=5, precise=ops.MODE_BF16)
    rel = ((dw16 - dw32).norm() / dw32.norm()).item()
    assert rel < 1e-2, rel
    # accumulate form and a different chunking give the same sums
    dw_acc = ops.conv3d_c1_wgrad_bn(x, y, dout, out, arg, mean, invstd, coef, pool, dw=dw16.clone(), beta=1, nchunk=3, precise=ops.MODE_BF16)
    close(dw_acc, (2 * dw16).cpu(), 1e-4, 1e-4 * dw16.abs().max().item())


@pytest.mark.parametrize("b,t,h,w", [(2, 3, 40, 24), (1, 2, 33, 50), (1, 9, 64, 80), (2, 8, 224, 224)])
def test_conv3d_c1_without_the_stored_conv_output_is_bit_identical(b, t, h, w):
    """The 16-bit first layer's three recompute passes (statistics only; conv -> BatchNorm -> 2x2 pool -> LeakyReLU; weight gradient
    with the tile's conv output recomputed) against the kernels that store y [B,T,H,W,16] and read it back: same partial sums, same
    pooled activation / IEEE-half copy / argmax bytes, same weight gradient -- bit for bit (odd sizes: partial tiles and a dropped
    last row / column of the pool; (1, 9, ...): a workgroup's tile walk ends inside a plane)."""
    from maavss_amd import ops
    x = torch.rand(b, t, h, w, generator=torch.Generator().manual_seed(11)).cuda()
    wgt = rnd(16, 1, 3, 5, 5, seed=12, scale=0.1).cuda()
    gamma, beta = (1 + 0.3 * rnd(16, seed=13)).cuda(), (0.2 * rnd(16, seed=14)).cuda()
    y, part = ops.conv3d_c1_fwd(x, wgt, want_stats=True, precise=ops.MODE_F16)
    y2, part2 = ops.conv3d_c1_stats(x, wgt, gamma)
    assert torch.equal(part, part2)
    mean, invstd = ops.bn_finalize(part, b * t * h * w)
    out, arg, out16 = ops.bn_pool_act_fwd(y, mean, invstd, gamma, beta, 2, ops.BN_LEAKY, want16=True)
    o2, a2, o16, ob = ops.conv3d_c1_bn_pool_act(x, wgt, mean, invstd, gamma, beta, want_bf16=True)
    assert ob.dtype == torch.bfloat16 and torch.equal(ob, o2.bfloat16())
    assert torch.equal(out, o2) and torch.equal(arg, a2) and torch.equal(out16.view(torch.int16), o16.view(torch.int16))
    dout = rnd(*out.shape, seed=15).cuda()
    dg, db = torch.zeros(16, device="cuda"), torch.zeros(16, device="cuda")
    coef = ops.bn_pool_act_bwd(dout, out, arg, y, mean, invstd, gamma, 2, ops.BN_LEAKY, dgamma=dg, dbeta=db, beta=beta, coef_only=True)
    dw = ops.conv3d_c1_wgrad_bn(x, y, dout, out, arg, mean, invstd, coef, 2, nchunk=5, precise=ops.MODE_BF16)
    dw2 = ops.conv3d_c1_wgrad_bn_recompute(x, wgt, dout, arg, mean, invstd, beta, coef, 2, nchunk=5)
    assert torch.equal(dw, dw2), (dw - dw2).abs().max().item()
    # the reduction of the BatchNorm backward does not touch y when every |gamma| >= 1e-2: hand it the uninitialised tensor
    dg2, db2 = torch.zeros(16, device="cuda"), torch.zeros(16, device="cuda")
    y2.fill_(float("nan"))
    coef2 = ops.bn_pool_act_bwd(dout, o2, a2, y2, mean, invstd, gamma, 2, ops.BN_LEAKY, dgamma=dg2, dbeta=db2, beta=beta, coef_only=True)
    assert torch.equal(coef, coef2) and torch.equal(dg, dg2) and torch.equal(db, db2)
    # degenerate channel (|gamma| < 1e-2): the statistics pass stores y after all, and the gather path of the reduction reads it
    gamma[5] = 1e-3
    y3, part3 = ops.conv3d_c1_stats(x, wgt, gamma)
    assert torch.equal(y3, y) and torch.equal(part3, part)


def test_bn_pool_strided_output():
    """last visual stage writes the [B,16,T,S] block of the LSTM sequence buffer directly"""
    from maavss_amd import ops
    b, t, c, h, w = 2, 4, 16, 6, 6
    y = rnd(b, t, h, w, c, seed=1)
    part = ops.bn_stats(y.cuda(), c)
    mean, invstd = ops.bn_finalize(part, b * t * h * w)
    ones, zeros = torch.ones(c).cuda(), torch.zeros(c).cuda()
    ref, _ = ops.bn_pool_act_fwd(y.cuda(), mean, invstd, ones, zeros, 3, 0)
    s = 4
    seq = torch.zeros(b, c, 2 * t * s, device="cuda")
    ops.bn_pool_act_fwd(y.cuda(), mean, invstd, ones, zeros, 3, 0, out=seq, strides=(c * 2 * t * s, s, 1, 2 * t * s))
    want = ref.permute(0, 4, 1, 2, 3).reshape(b, c, t * s)
    assert torch.equal(seq[:, :, :t * s], want) and seq[:, :, t * s:].abs().max().item() == 0


# ----------------------------------------------------------------------------------------------- conv2d
@pytest.mark.parametrize("ci,co,stride,pw,h,w,nchw", [(2, 4, (2, 2), 3, 16, 33, True), (4, 8, (2, 2), 4, 8, 16, False),
                                                       (16, 16, (1, 2), 4, 8, 32, False), (2, 4, (2, 2), 3, 64, 257, True)])
def test_conv2d(ci, co, stride, pw, h, w, nchw):
    from maavss_amd import ops
    b = 2
    x = rnd(b, ci, h, w, seed=1).requires_grad_(True)
    wgt = rnd(co, ci, 3, 9, seed=2, scale=0.2).requires_grad_(True)
    y_ref = F.conv2d(x, wgt, stride=stride, padding=(1, pw))
    dy = rnd(*y_ref.shape, seed=3)
    gx, gw = torch.autograd.grad(y_ref, (x, wgt), dy)
    xin = x.detach().cuda() if nchw else x.detach().permute(0, 2, 3, 1).contiguous().cuda()
    y = ops.conv2d_fwd(xin, wgt.detach().cuda(), stride, pw, nchw)
    close(y.permute(0, 3, 1, 2), y_ref, 1e-5, 1e-5)
    dyc = dy.permute(0, 2, 3, 1).contiguous().cuda()
    dx = ops.conv2d_dgrad(dyc, wgt.detach().cuda(), (h, w), stride, pw)
    close(dx.permute(0, 3, 1, 2), gx, 1e-5, 1e-5)
    dw = ops.conv2d_wgrad(xin, dyc, wgt.shape, stride, pw, nchw)
    close(dw, gw, 1e-4, 1e-4 * gw.abs().max().item())


# ----------------------------------------------------------------------------------------------- LSTM
@pytest.mark.parametrize("b", [2, 37])
def test_lstm(b):
    from maavss_amd import ops
    l, n_in = 16, 64
    lstm = torch.nn.LSTM(n_in, 256, 1, bias=False, batch_first=True, bidirectional=True)
    x = rnd(b, l, n_in, seed=1).requires_grad_(True)
    out_ref, _ = lstm(x)
    dout = rnd(*out_ref.shape, seed=2)
    grads = torch.autograd.grad(out_ref, [x] + list(lstm.parameters()), dout)
    p = {k: v.detach() for k, v in lstm.named_parameters()}
    wih = torch.cat([p["weight_ih_l0"], p["weight_ih_l0_reverse"]], 0).cuda()
    whf, whb = p["weight_hh_l0"].cuda(), p["weight_hh_l0_reverse"].cuda()
    xc = x.detach().cuda().reshape(b * l, n_in)
    gx = ops.gemm(xc, wih, precise=True).reshape(b, l, 2, 4, 256)
    av, hp, gs, cs = ops.lstm_fwd(gx, whf, whb)
    close(av, out_ref, 1e-4, 1e-5)
    dgx = ops.lstm_bwd(dout.cuda(), whf, whb, gs, cs).reshape(b * l, 2048)
    dx = ops.gemm(dgx, wih, trans_b=True, precise=True).reshape(b, l, n_in)
    close(dx, grads[0], 1e-3, 2e-5)
    dwih = ops.gemm(dgx, xc, trans_a=True, trans_b=True, precise=True)
    close(dwih[:1024], grads[1], 1e-3, 1e-4)
    close(dwih[1024:], grads[3], 1e-3, 1e-4)
    hp2 = hp.reshape(b * l, 512)
    dwhf = ops.gemm(dgx[:, :1024], hp2[:, :256], trans_a=True, trans_b=True, precise=True)
    dwhb = ops.gemm(dgx[:, 1024:], hp2[:, 256:], trans_a=True, trans_b=True, precise=True)
    close(dwhf, grads[2], 1e-3, 1e-4)
    close(dwhb, grads[4], 1e-3, 1e-4)


# ----------------------------------------------------------------------------------------------- loss / Adam
def test_mse_pair_and_act_bwd():
    from maavss_amd import ops
    a, ta = rnd(4, 2, 8, 257, seed=1).requires_grad_(True), rnd(4, 2, 8, 257, seed=2)
    v, tv = torch.rand(4, 1, 64, 64, generator=torch.Generator().manual_seed(3)).requires_grad_(True), rnd(4, 1, 64, 64, seed=4)
    la, lv = F.mse_loss(a, ta), F.mse_loss(v, tv)
    tot = (la + 0.001 * lv) / 4
    ga, gv = torch.autograd.grad(tot, (a, v))
    losses, d_a, d_v = ops.mse_pair(a.detach().cuda(), ta.cuda(), v.detach().cuda(), tv.cuda(), 0.001, 4)
    close(losses, torch.stack([la, lv, tot]), 1e-5, 1e-7)
    close(d_a, ga, 1e-5, 1e-9)
    close(d_v, gv, 1e-5, 1e-10)
    z = rnd(1000, seed=5).requires_grad_(True)
    for act, fn in ((1, torch.tanh), (2, torch.sigmoid)):
        o = fn(z)
        g, = torch.autograd.grad(o, z, torch.ones_like(o) * 0.5)
        close(ops.act_bwd(torch.full((1000,), 0.5).cuda(), o.detach().cuda(), act), g, 1e-5, 1e-7)


def test_adam_matches_torch():
    from maavss_amd import ops
    n = 100003
    p = torch.nn.Parameter(rnd(n, seed=1))
    opt = torch.optim.Adam([p], lr=1e-3)
    pc, m, v = p.detach().clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda()
    for step in range(1, 4):
        g = rnd(n, seed=10 + step)
        p.grad = g.clone()
        opt.step()
        ops.adam_step(pc, g.cuda(), m, v, 1e-3, step)
    close(pc, p.detach(), 1e-5, 1e-6)


@pytest.mark.parametrize("ci,co,kw,stride,opad,nhwc", [(16, 8, 9, (2, 2), (1, 1), True), (4, 4, 9, (2, 2), (1, 1), True),
                                                     (4, 2, 10, (1, 2), (0, 1), False), (8, 4, 9, (2, 1), (1, 0), True)])
def test_convt2d_fwd_dgrad_wgrad_vs_torch(ci, co, kw, stride, opad, nhwc):
    """K11: ConvTranspose2d(k=(3,kw), padding (1,4)) forward / input gradient / weight gradient vs torch on the CPU."""
    import torch.nn.functional as F
    from maavss_amd import ops
    g = torch.Generator().manual_seed(5)
    b, hi, wi = 2, 9, 17
    x = torch.randn(b, ci, hi, wi, generator=g)
    w = torch.randn(ci, co, 3, kw, generator=g) * 0.2
    x.requires_grad_(True)
    w.requires_grad_(True)
    y = F.conv_transpose2d(x, w, stride=stride, padding=(1, 4), output_padding=opad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    x_nhwc = x.detach().permute(0, 2, 3, 1).contiguous().cuda()
    wc = w.detach().cuda()
    got = ops.convt2d_fwd(x_nhwc, wc, stride, opad, out_nhwc=nhwc)
    got_nchw = got.permute(0, 3, 1, 2) if nhwc else got
    assert tuple(got_nchw.shape) == tuple(y.shape)
    np.testing.assert_allclose(got_nchw.cpu().numpy(), y.detach().numpy(), rtol=1e-5, atol=1e-5)
    dyc = (dy.permute(0, 2, 3, 1) if nhwc else dy).contiguous().cuda()
    dx = ops.convt2d_dgrad(dyc, wc, (hi, wi), stride, opad, out_nhwc=nhwc)
    np.testing.assert_allclose(dx.permute(0, 3, 1, 2).cpu().numpy(), x.grad.numpy(), rtol=1e-5, atol=2e-5)
    dw = ops.convt2d_wgrad(x_nhwc, dyc, w.shape, stride, opad, out_nhwc=nhwc)
    np.testing.assert_allclose(dw.cpu().numpy(), w.grad.numpy(), rtol=1e-4, atol=1e-4)
