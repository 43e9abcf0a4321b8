"""Round-2 parity gates ("Next round" items 1 and 8), all through the C-ABI:

  * the configuration bench.py actually runs -- T = 16, 224^2, spatial_match="adaptive", default 16-bit conv modes --
    against the fp32 oracle twin: mask-MSE, loss, and per-parameter gradient SAMPLES;
  * end to end with the ViT in the loop: frames -> VideoAttention (bf16 HIP) -> AV_Fusion_Model_Frames (HIP) against
    vit_ref_cpu -> clip_normalise_ref -> AVFusionFramesRef in fp32 (av_dataset.py:321-333 -> train_avse_frames.py:164-168);
  * the ViT against an oracle that rounds to bf16 where the kernels do (tight), the fp32 distance reported next to it;
  * grad toggles and av_fusion_forward of the drop-in model.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# Operand rounding of the 16-bit paths: forward operands are IEEE half (unit roundoff 2^-11), backward operands bf16
# (2^-8 per element, 2^-9 rms).  A gradient element is a sum of products of two rounded operands, chained through at
# most 5 conv layers whose backward inputs are themselves rounded results: |err| <= ~ L * sqrt(2) * 2^-9 * (|g| + rms(g))
# with L = 5 -> 1.4e-2; the tests allow 2.5e-2 (x1.8 margin) per sampled element and 1.5e-2 on the tensor norms.
BF16_GRAD_RTOL = 2.5e-2


def _build(batch, frames, width, fft_len, seed, precise, spatial_match):
    import maavss_amd
    from oracle import avse_ref_cpu as orc
    hpf = 8
    n_bins, t_a = fft_len // 2 + 1, hpf * frames
    shapes = ([batch, 2, t_a, n_bins], [batch, 1, frames, width, width], hpf)
    model = maavss_amd.AV_Fusion_Model_Frames(*shapes, precise=precise, spatial_match=spatial_match)
    twin = orc.AVFusionFramesRef(*shapes, spatial_match=spatial_match)
    model.load_state_dict(orc.seeded_state_dict(twin, seed), strict=True)
    orc.load_seeded(twin, seed)
    return model.to("cuda").train(), twin.train(), orc.synthetic_batch(batch, frames, width, t_a, n_bins, hpf, seed + 1)


def test_benched_configuration_16bit_against_fp32_oracle():
    """BASELINE config[1] as bench.py runs it (B reduced to 2 for the CPU oracle): T=16, 224^2, adaptive, 16-bit modes."""
    from oracle import avse_ref_cpu as orc
    model, twin, (x_a, x_v, y_a, y_v) = _build(2, 16, 224, 512, 41, precise=False, spatial_match="adaptive")
    loss_ref, a_loss_ref, v_loss_ref, (a_ref, v_ref, f_ref) = orc.loss_ref(twin, x_a, x_v, y_a, y_v, 0.001, 1)
    loss_ref.backward()
    a, v, fused = model(x_a.cuda(), x_v.cuda())
    loss = F.mse_loss(a, y_a.cuda()) + 0.001 * F.mse_loss(v, y_v.cuda())
    loss.backward()
    mse = float(((a.detach().cpu() - a_ref.detach()) ** 2).mean())
    print(f"[parity] benched config: mask-MSE {mse:.3e}, |dloss| {abs(loss.item() - loss_ref.item()):.3e}")
    assert mse <= 1e-5, mse                                             # BASELINE.json: mask MSE within 1e-5
    assert abs(loss.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item()) + 1e-5
    assert (v.detach().cpu() - v_ref.detach()).abs().max().item() < 2e-3
    ref = dict(twin.named_parameters())
    worst = 0.0
    for k, p in model.named_parameters():
        if k.startswith("stft_autoencoder.") or ref[k].grad is None:
            continue
        g, gr = p.grad.detach().cpu().flatten(), ref[k].grad.flatten()
        rms = gr.norm().item() / np.sqrt(gr.numel())
        assert abs(g.norm().item() - gr.norm().item()) <= 1.5e-2 * gr.norm().item() + 1e-9, k
        idx = torch.randperm(gr.numel(), generator=torch.Generator().manual_seed(7))[:256]
        err = (g[idx] - gr[idx]).abs()
        bound = BF16_GRAD_RTOL * (gr[idx].abs() + rms)
        worst = max(worst, float((err / bound).max()))
        assert bool((err <= bound).all()), (k, float((err / bound).max()))
    print(f"[parity] benched config: worst sampled gradient error = {worst:.2f} of the operand-rounding bound")


def test_end_to_end_frames_to_mask_with_the_vit_in_the_loop():
    """frames -> attention frames (bf16 HIP ViT) -> AVSE (16-bit HIP) vs the all-fp32 oracle chain on the pinned P shape."""
    import maavss_amd
    from oracle import avse_ref_cpu as orc, vit_ref_cpu as vref
    b, t, w, hpf = 2, 8, 256, 8
    model, twin, (x_a, _, y_a, _) = _build(b, t, w, 512, 43, precise=False, spatial_match="exact")
    sd = vref.seeded_vit_state(3)
    va = maavss_amd.VideoAttention(path_to_weights="/nonexistent.pth")
    va.load_state_dict(sd)
    frames = vref.synthetic_frames(b * t, w, 9)
    with torch.no_grad():
        x_v_ref = torch.stack([vref.clip_normalise_ref(vref.inference_ref(sd, frames[i * t:(i + 1) * t])) for i in range(b)])
    x_v = va.attention_frames(frames.cuda(), clip_frames=t).view(b, 1, t, w, w)
    y_v_ref, y_v = x_v_ref[:, :, t // 2], x_v[:, :, t // 2]
    loss_ref, _, _, (a_ref, v_ref, _) = orc.loss_ref(twin, x_a, x_v_ref, y_a, y_v_ref, 0.001, 1)
    a, v, fused = model(x_a.cuda(), x_v)
    loss = F.mse_loss(a, y_a.cuda()) + 0.001 * F.mse_loss(v, y_v)
    map_err = (x_v.cpu() - x_v_ref).abs()
    mse = float(((a.detach().cpu() - a_ref.detach()) ** 2).mean())
    print(f"[parity] end to end: attention maps max|err| {map_err.max().item():.3e} mean {map_err.mean().item():.3e}; "
          f"mask-MSE {mse:.3e}; |dloss| {abs(loss.item() - loss_ref.item()):.3e} (loss {loss_ref.item():.5f})")
    assert mse <= 1e-5, mse
    assert abs(loss.item() - loss_ref.item()) <= 1e-4


@pytest.mark.parametrize("width,frames", [(64, 4), (224, 2)])
def test_video_attention_matches_bf16_emulating_oracle(width, frames):
    """Against an oracle that rounds to bf16 exactly where the kernels store 16-bit values: what remains is summation
    order, the deferred running maximum and the polynomial GELU -- an order of magnitude below the quantisation error
    itself, so a wrong position-embedding row, LayerNorm eps or softmax scale cannot hide in it."""
    import maavss_amd
    from oracle import vit_ref_cpu as vref
    sd = vref.seeded_vit_state(3)
    va = maavss_amd.VideoAttention(path_to_weights="/nonexistent.pth")
    va.load_state_dict(sd)
    fr = vref.synthetic_frames(frames, width, 5)
    with torch.no_grad():
        want_emu = vref.inference_ref(sd, fr, emulate_bf16=True)
        want_f32 = vref.inference_ref(sd, fr)
        cls_emu = vref.cls_attention(sd, fr, emulate_bf16=True)
    got = va._inference(fr)
    got_cls = va.cls_attention(fr.cuda()).cpu()
    e_emu, e_f32 = (got - want_emu).abs().max().item(), (got - want_f32).abs().max().item()
    print(f"[parity] ViT {width}^2: maps max|err| vs bf16-emulating oracle {e_emu:.3e} (mean {(got - want_emu).abs().mean().item():.2e}); "
          f"vs fp32 oracle {e_f32:.3e} = quantisation error")
    assert e_emu <= 5e-3, e_emu
    assert (got - want_emu).abs().mean().item() <= 5e-4
    rel = (got_cls - cls_emu).abs().max().item() / cls_emu.abs().max().item()
    assert rel <= 5e-3, rel


def test_grad_toggles_on_the_frames_model(golden_dir):
    """toggle_enc_grads / toggle_fusion_grads (avse_model_final.py:216-232): frozen parameters get no gradient, the
    others keep exactly the gradient of the unfrozen run (golden S)."""
    import os
    z = np.load(os.path.join(golden_dir, "avse_S.npz"), allow_pickle=False)
    m = {k[5:]: z[k].item() for k in z.files if k.startswith("meta_")}
    names = [str(k) for k in z["param_names"]]
    norm = dict(zip(names, z["grad_norm"]))
    enc = lambda n: n.startswith("visual_encoder.") or n.startswith("stft_encoder.")      # noqa: E731
    fusion = lambda n: n.split(".")[0] in ("lstm", "fc1", "fc2", "a_fc1", "v_fc1")         # noqa: E731
    for toggle, frozen in (("toggle_enc_grads", enc), ("toggle_fusion_grads", fusion)):
        model, _, (x_a, x_v, y_a, y_v) = _build(m["batch"], m["frames"], m["width"], m["fft_len"], m["seed"], True, "exact")
        getattr(model, toggle)(False)
        a, v, _ = model(x_a.cuda(), x_v.cuda())
        (F.mse_loss(a, y_a.cuda()) + m["loss_coeff"] * F.mse_loss(v, y_v.cuda())).backward()
        for n, p in model.named_parameters():
            if n.startswith("stft_autoencoder.") or n.startswith("stft_decoder."):
                continue
            if frozen(n):
                assert p.grad is None, (toggle, n)
            else:
                gn = p.grad.double().norm().item()
                assert abs(gn - norm[n]) <= 2e-3 * norm[n] + 1e-7, (toggle, n, gn, norm[n])
        getattr(model, toggle)(True)
        assert all(p.requires_grad for n, p in model.named_parameters() if frozen(n))
    # TrainStep re-reads requires_grad at every call: freeze after construction, gradients stay zero and
    # Adam leaves the frozen weights alone
    import maavss_amd
    model, _, (x_a, x_v, y_a, y_v) = _build(m["batch"], m["frames"], m["width"], m["fft_len"], m["seed"], True, "exact")
    step = maavss_amd.TrainStep(model, lr=1e-3)
    model.toggle_enc_grads(False)
    w_enc, w_fc = model.visual_encoder[0].weight.detach().clone(), model.fc2.weight.detach().clone()
    step(x_a.cuda(), x_v.cuda(), y_a.cuda(), y_v.cuda())
    assert torch.equal(model.visual_encoder[0].weight.detach(), w_enc)
    assert not torch.equal(model.fc2.weight.detach(), w_fc)
    assert step.opt.steps["visual_encoder.0.weight"] == 0 and step.opt.steps["fc2.weight"] == 1


def test_av_fusion_forward_from_given_encodings():
    """av_fusion_forward(x_a_enc, x_v_enc) (avse_model_final.py:235-251) as a public entry: value and all gradients."""
    model, twin, _ = _build(2, 8, 128, 256, 47, True, "exact")
    g = torch.Generator().manual_seed(3)
    xa = (torch.rand(2, 16, 8, 4, generator=g) * 2 - 1).requires_grad_()
    xv = torch.rand(2, 16, 8, 4, generator=g).requires_grad_()
    ref = twin.av_fusion_forward(xa, xv)
    w = torch.randn(ref.shape, generator=g)
    (ref * w).sum().backward()
    xa_c, xv_c = xa.detach().cuda().requires_grad_(), xv.detach().cuda().requires_grad_()
    got = model.av_fusion_forward(xa_c, xv_c)
    assert tuple(got.shape) == (2, 512)
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=2e-5)
    (got * w.cuda()).sum().backward()
    np.testing.assert_allclose(xa_c.grad.cpu().numpy(), xa.grad.numpy(), rtol=2e-3, atol=1e-6)
    np.testing.assert_allclose(xv_c.grad.cpu().numpy(), xv.grad.numpy(), rtol=2e-3, atol=1e-6)
    ref_p = dict(twin.named_parameters())
    for n, p in model.named_parameters():
        if n.split(".")[0] in ("lstm", "fc1", "fc2"):
            gr = ref_p[n].grad
            assert (p.grad.cpu() - gr).norm().item() <= 2e-3 * gr.norm().item() + 1e-8, n
        elif not n.startswith("stft_autoencoder."):
            assert p.grad is None, n
    with pytest.raises(ValueError):
        model.av_fusion_forward(xa_c[:, :, :4], xv_c)
