"""GPU parity of the phasegram variant `AV_Fusion_Model` (avse_model.py:410-711) against the numbers
the reference's own class produced (tests/golden/avfm_A.npz, oracle/make_golden.py) -- through the C-ABI -- and of the
generic convolution kernels K19 against torch on the CPU."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "avfm_A.npz"), allow_pickle=False)
    return z, {k[5:]: z[k].item() for k in z.files if k.startswith("meta_")}


def _setup(golden_dir):
    import maavss_amd
    from oracle import avfm_ref_cpu as avfm
    z, m = _golden(golden_dir)
    b, t_a, n_bins, t, p = m["batch"], m["t_a"], m["n_bins"], m["frames"], m["p_size"]
    stft_shape, pgram_shape = [b, 2, t_a, n_bins], [b, 1, t, p * p]
    model = maavss_amd.AV_Fusion_Model(stft_shape, pgram_shape, 8)
    twin = avfm.AVFusionRef(stft_shape, pgram_shape, 8)
    assert list(model.state_dict().keys()) == list(twin.state_dict().keys())
    model.load_state_dict(avfm.seeded_state_dict(twin, m["seed"]), strict=True)
    g = torch.Generator().manual_seed(m["seed"] + 5)
    attn = torch.rand(b, 1, t, p, p, generator=g)
    np.testing.assert_array_equal(attn.flatten()[::97].numpy(), z["attn_sample"])       # same inputs as the fixture
    x_v = avfm.video_phasegram_ref(attn)
    x_a = torch.randn(b, 2, t_a, n_bins, generator=g) * 0.5
    y_a = torch.randn(b, 2, t_a, n_bins, generator=g) * 0.3
    return z, model.cuda(), x_a, x_v, y_a


def _check_grads(z, tag, model, rtol_norm=2e-3):
    params = dict(model.named_parameters())
    for i, k in enumerate(z[f"{tag}_param_names"]):
        k = str(k)
        g = params[k].grad
        ref_n = z[f"{tag}_grad_norm"][i]
        assert g is not None, k
        gn = g.double().norm().item()
        if ref_n < 1e-4:
            # a convolution bias in front of a train-mode BatchNorm has a mathematically zero gradient: both sides hold
            # rounding noise (1e-6) there, nothing to compare
            assert gn < 1e-4, (k, gn, ref_n)
            continue
        assert abs(gn - ref_n) <= rtol_norm * ref_n + 1e-7, (k, gn, ref_n)
        flat = g.flatten()
        idx = (torch.arange(8) * (flat.numel() - 1)) // 7
        scale = ref_n / np.sqrt(flat.numel())
        np.testing.assert_allclose(flat[idx.cuda()].cpu().numpy(), z[f"{tag}_grad_sample"][i], rtol=5e-3, atol=5e-2 * scale, err_msg=k)


def test_forward_backward_matches_reference_golden(golden_dir):
    z, model, x_a, x_v, y_a = _setup(golden_dir)
    model.train()
    xa, xv = x_a.cuda(), x_v.cuda()
    yh_a, yh_v, fused = model(xa, xv)
    assert yh_a.shape == xa.shape and yh_v.shape == xv.shape and tuple(fused.shape) == (2, 512)
    loss = F.mse_loss(yh_v, xv) + F.mse_loss(yh_a, y_a.cuda())
    loss.backward()
    np.testing.assert_allclose(fused.detach().cpu().numpy(), z["full_fused"], rtol=0, atol=3e-5)
    np.testing.assert_allclose(yh_a.detach().flatten()[::127].cpu().numpy(), z["full_a_sample"], rtol=0, atol=3e-5)
    np.testing.assert_allclose(yh_v.detach().flatten()[::31].cpu().numpy(), z["full_v_sample"], rtol=0, atol=3e-5)
    assert abs(loss.item() - z["full_loss"]) < 2e-6
    _check_grads(z, "full", model)
    bufs = dict(model.named_buffers())
    for i, k in enumerate(z["full_bn_names"]):
        k = str(k)
        if k.startswith(("phasegram_decoder.", "stft_decoder.")):
            continue                                    # untouched by forward(); the fixture saw them after the AE steps
        assert abs(bufs[k + ".running_mean"].double().sum().item() - z["full_bn_running_mean_sum"][i]) < 1e-4, k
        assert abs(bufs[k + ".running_var"].double().sum().item() - z["full_bn_running_var_sum"][i]) < 1e-3, k
    # the decoders take no part in forward(): no gradient, like the reference
    assert all(p.grad is None for n, p in model.named_parameters() if n.startswith(("phasegram_decoder.", "stft_decoder.")))


def test_frozen_encoders_as_in_train_av_net(golden_dir):
    """train_av_net.py:73-75,97-100: autoencoder parameters frozen, only the fusion network trains -- its gradients are
    the same numbers, the encoders get none (and their backward kernels are skipped)."""
    z, model, x_a, x_v, y_a = _setup(golden_dir)
    model.train()
    model.toggle_phasegram_ae_grads(False)
    model.toggle_stft_ae_grads(False)
    xa, xv = x_a.cuda(), x_v.cuda()
    yh_a, yh_v, _ = model(xa, xv)
    (F.mse_loss(yh_v, xv) + F.mse_loss(yh_a, y_a.cuda())).backward()
    params = dict(model.named_parameters())
    for i, k in enumerate(z["full_param_names"]):
        k = str(k)
        if k.startswith(("phasegram_encoder.", "stft_encoder.")):
            assert params[k].grad is None, k
        else:
            gn = params[k].grad.double().norm().item()
            assert abs(gn - z["full_grad_norm"][i]) <= 2e-3 * z["full_grad_norm"][i] + 1e-6, k


@pytest.mark.parametrize("tag", ["vae", "aae"])
def test_autoencoder_entry_points_match_reference_golden(golden_dir, tag):
    z, model, x_a, x_v, _ = _setup(golden_dir)
    model.train()
    # the fixture ran forward() first (BatchNorm buffers of the encoders moved once): replay that, then the AE step
    model(x_a.cuda(), x_v.cuda())
    if tag == "aae":
        model.visual_ae_forward(x_v.cuda())
    x = (x_v if tag == "vae" else x_a).cuda()
    yh = model.visual_ae_forward(x) if tag == "vae" else model.audio_ae_forward(x)
    assert yh.shape == x.shape
    loss = F.mse_loss(yh, x)
    loss.backward()
    np.testing.assert_allclose(yh.detach().flatten()[::53].cpu().numpy(), z[f"{tag}_out_sample"], rtol=0, atol=3e-5)
    assert abs(loss.item() - z[f"{tag}_loss"]) < 2e-6
    _check_grads(z, tag, model)


def test_constructor_guard_and_eval_mode(golden_dir):
    import maavss_amd
    from oracle import avfm_ref_cpu as avfm
    with pytest.raises(ValueError):
        maavss_amd.AV_Fusion_Model([2, 2, 32, 128], [2, 1, 4, 4096], 8)          # LSTM over 4 rows != fc_size / 512
    z, model, x_a, x_v, _ = _setup(golden_dir)
    m = {k[5:]: z[k].item() for k in z.files if k.startswith("meta_")}
    twin = avfm.AVFusionRef([2, 2, m["t_a"], m["n_bins"]], [2, 1, m["frames"], m["p_size"] ** 2], 8)
    avfm.load_seeded(twin, m["seed"])
    twin.eval()
    model.eval()
    with torch.no_grad():
        ref = twin(x_a, x_v)
        got = model(x_a.cuda(), x_v.cuda())
    for r, g_ in zip(ref, got):
        np.testing.assert_allclose(g_.cpu().numpy(), r.numpy(), rtol=0, atol=5e-5)


@pytest.mark.parametrize("kind,ci,co,k,stride,pad,opad", [("conv", 1, 2, (1, 9), (1, 2), (0, 4), None), ("conv", 8, 32, (5, 5), (2, 2), (2, 2), None),
                                                         ("conv", 4, 8, (5, 5), (1, 2), (2, 2), None), ("convT", 16, 8, (1, 9), (1, 2), (0, 4), (0, 1)),
                                                         ("convT", 16, 4, (5, 5), (2, 2), (2, 2), (1, 1)), ("convT", 2, 2, (5, 5), (2, 1), (2, 2), (1, 0))])
def test_generic_conv_kernels_vs_torch(kind, ci, co, k, stride, pad, opad):
    from maavss_amd import ops
    g = torch.Generator().manual_seed(3)
    b, hi, wi = 2, 6, 20
    x = torch.randn(b, ci, hi, wi, generator=g, requires_grad=True)
    if kind == "conv":
        w = (torch.randn(co, ci, *k, generator=g) * 0.2).requires_grad_(True)
        bias = torch.randn(co, generator=g).requires_grad_(True)
        y = F.conv2d(x, w, bias, stride, pad)
    else:
        w = (torch.randn(ci, co, *k, generator=g) * 0.2).requires_grad_(True)
        bias = torch.randn(co, generator=g).requires_grad_(True)
        y = F.conv_transpose2d(x, w, bias, stride, pad, opad)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xm = ops.Map(x.detach().cuda(), nchw=True)                                   # NCHW in, channels-last (padded) out
    cp = max(co, 4)
    yt = torch.zeros(b, y.shape[2], y.shape[3], cp, device="cuda")
    ym = ops.Map(yt, c=co)
    wc, bc = w.detach().cuda(), bias.detach().cuda()
    if kind == "conv":
        ops.conv_gen_small(xm, wc, bc, ym, stride, pad)          # big = x, small = y
    else:
        ops.conv_gen_big(xm, wc, bc, ym, stride, pad)            # small = x, big = y
    np.testing.assert_allclose(yt[..., :co].permute(0, 3, 1, 2).cpu().numpy(), y.detach().numpy(), rtol=1e-5, atol=2e-5)
    if cp > co:
        assert float(yt[..., co:].abs().max()) == 0.0            # dead channels untouched
    dyt = torch.zeros_like(yt)
    dyt[..., :co] = dy.permute(0, 2, 3, 1).cuda()
    dym = ops.Map(dyt, c=co)
    dxt = torch.empty_like(xm.t)
    dxm = ops.Map(dxt, nchw=True)
    if kind == "conv":
        ops.conv_gen_big(dym, wc, None, dxm, stride, pad)
        dw = ops.conv_gen_wgrad(dym, xm, w.shape, stride, pad)
    else:
        ops.conv_gen_small(dym, wc, None, dxm, stride, pad)
        dw = ops.conv_gen_wgrad(xm, dym, w.shape, stride, pad)
    np.testing.assert_allclose(dxt.cpu().numpy(), x.grad.numpy(), rtol=1e-5, atol=3e-5)
    np.testing.assert_allclose(dw.cpu().numpy(), w.grad.numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(ops.channel_sum(dym).cpu().numpy(), bias.grad.numpy(), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("p,diff,cumulative,normalize", [(32, True, True, True), (64, True, True, True), (64, False, True, True),
                                                        (32, True, False, False), (64, False, False, True)])
def test_video_phasegram_matches_oracle(golden_dir, p, diff, cumulative, normalize):
    """K20 vs the oracle restatement of utilities.video_phasegram (itself checked against the reference function, and against
    the committed samples of its output below)."""
    import maavss_amd
    from oracle import avfm_ref_cpu as avfm
    g = torch.Generator().manual_seed(7 + 5)          # the fixture's inputs (seed + 5)
    attn = torch.rand(2, 1, 8, p, p, generator=g)
    want = avfm.video_phasegram_ref(attn, diff=diff, cumulative=cumulative, normalize=normalize)
    got = maavss_amd.video_phasegram(attn.cuda(), resize=(p, p), diff=diff, cumulative=cumulative, normalize=normalize)
    assert tuple(got.shape) == (2, 1, 8, p * p)
    # the phase of a bin is ill-conditioned where the bin is small (float FFT noise / |X|); after the cumulative sum and the
    # normalisation that is a few 1e-6 of full scale
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=2e-5 * float(want.abs().max()))
    if (p, diff, cumulative, normalize) == (32, True, True, True):
        z = np.load(os.path.join(golden_dir, "avfm_A.npz"), allow_pickle=False)
        np.testing.assert_array_equal(attn.flatten()[::97].numpy(), z["attn_sample"])
        np.testing.assert_allclose(got.flatten()[::13].cpu().numpy(), z["pgram_sample"], rtol=0, atol=2e-5)


def test_video_phasegram_with_resize():
    """resize=(p, p) as train_av_net.py:122-125 calls it: torchvision's tensor resize = F.interpolate(bilinear,
    align_corners=False); checked against torch on the CPU, then the phasegram of the resized frames against the oracle."""
    import maavss_amd
    from maavss_amd import _lib
    from oracle import avfm_ref_cpu as avfm
    g = torch.Generator().manual_seed(3)
    attn = torch.rand(2, 1, 4, 224, 224, generator=g)
    small_ref = F.interpolate(attn[:, 0], size=(64, 64), mode="bilinear", align_corners=False).unsqueeze(1)
    xc = attn.cuda()
    small = torch.empty(2, 1, 4, 64, 64, device="cuda")
    _lib.call("maavss_resize_bilinear", _lib.ptr(xc), _lib.ptr(small), 8, 224, 224, 64, 64, _lib.stream_ptr())
    np.testing.assert_allclose(small.cpu().numpy(), small_ref.numpy(), rtol=0, atol=2e-6)
    got = maavss_amd.video_phasegram(xc, resize=(64, 64))
    want = avfm.video_phasegram_ref(small_ref)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=5e-5 * float(want.abs().max()))


def test_av_fusion_forward_from_given_encodings(golden_dir):
    """AV_Fusion_Model.av_fusion_forward(x_a_enc, x_v_enc) (avse_model.py:658-670) as a public entry: value
    and every gradient (both encodings, LSTM / fc weights and biases) against the oracle twin."""
    from oracle import avfm_ref_cpu as avfm
    z, model, _, _, _ = _setup(golden_dir)
    m = {k[5:]: z[k].item() for k in z.files if k.startswith("meta_")}
    b = m["batch"]
    twin = avfm.AVFusionRef([b, 2, m["t_a"], m["n_bins"]], [b, 1, m["frames"], m["p_size"] ** 2], 8)
    avfm.load_seeded(twin, m["seed"])
    g = torch.Generator().manual_seed(11)
    xa = (torch.rand(b, model.c_a, model.h, model.w_enc, generator=g) * 2 - 1).requires_grad_()
    xv = (torch.rand(b, model.c_v, model.h, model.w_enc, generator=g) * 2 - 1).requires_grad_()
    ref = twin.av_fusion_forward(xa, xv)
    w = torch.randn(ref.shape, generator=g)
    (ref * w).sum().backward()
    xa_c, xv_c = xa.detach().cuda().requires_grad_(), xv.detach().cuda().requires_grad_()
    got = model.av_fusion_forward(xa_c, xv_c)
    assert tuple(got.shape) == (b, 512)
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=3e-5)
    (got * w.cuda()).sum().backward()
    np.testing.assert_allclose(xa_c.grad.cpu().numpy(), xa.grad.numpy(), rtol=2e-3, atol=2e-6)
    np.testing.assert_allclose(xv_c.grad.cpu().numpy(), xv.grad.numpy(), rtol=2e-3, atol=2e-6)
    ref_p = dict(twin.named_parameters())
    for n, p in model.named_parameters():
        if n.split(".")[0] in ("lstm", "fc1", "fc2"):
            gr = ref_p[n].grad
            assert (p.grad.cpu() - gr).norm().item() <= 2e-3 * gr.norm().item() + 1e-8, n
        elif not n.startswith(("stft_autoencoder.", "phasegram_autoencoder.")):
            assert p.grad is None, n
    with pytest.raises(ValueError):
        model.av_fusion_forward(xa_c[:, :, :2], xv_c)


def test_backward_through_an_eval_mode_forward(golden_dir):
    """torch autograd allows loss.backward() after model.eval() (BatchNorm with running statistics, no batch-mean terms in its
    backward): VERDICT r2 'missing' item 6 -- the phasegram model against its oracle twin in eval()."""
    from oracle import avfm_ref_cpu as avfm
    z, model, x_a, x_v, y_a = _setup(golden_dir)
    m = {k[5:]: z[k].item() for k in z.files if k.startswith("meta_")}
    twin = avfm.AVFusionRef([m["batch"], 2, m["t_a"], m["n_bins"]], [m["batch"], 1, m["frames"], m["p_size"] ** 2], 8)
    avfm.load_seeded(twin, m["seed"])
    twin.eval()
    model.eval()
    a_r, v_r, _ = twin(x_a, x_v)
    (F.mse_loss(a_r, y_a) + 0.5 * F.mse_loss(v_r, x_v)).backward()
    a, v, _ = model(x_a.cuda(), x_v.cuda())
    (F.mse_loss(a, y_a.cuda()) + 0.5 * F.mse_loss(v, x_v.cuda())).backward()
    ref_p = dict(twin.named_parameters())
    for n, p in model.named_parameters():
        gr = ref_p[n].grad
        if gr is None:
            continue
        assert p.grad is not None, n
        assert (p.grad.cpu() - gr).norm().item() <= 3e-3 * gr.norm().item() + 1e-7, (n, (p.grad.cpu() - gr).norm().item(), gr.norm().item())
