"""Parity at BASELINE config[1]'s FULL size (32 clips x 16 frames x 224^2, 512-pt STFT -- what bench.py runs), where the CPU oracle
cannot be run on the whole batch in test time: size-independent properties tie the full-size results to small cases that ARE checked
against the oracle, in this file and in the per-stage suites.

  * extractor: the attention frames of clip c inside the 512-frame batch are BIT-IDENTICAL to the same clip extracted alone (rows,
    (frame, head) pairs and clips are independent units), and the clips checked against the fp32 oracle chain sit inside the batch;
  * STFT: a sample inside the batch of 32 is bit-identical to the same sample transformed alone; exact homogeneity under a power of
    two, additivity to rounding; oracle on samples of the batch;
  * fusion network, eval mode (BatchNorm on running statistics: clips are independent): outputs of the batch of 32 equal the outputs
    of its pairs, the parameter gradients of the batch equal the SUM of its pairs' gradients (additivity over clips), one pair checked
    against the oracle twin forward and backward;
  * fusion network, train mode (BatchNorm couples the clips): the backward pass is LINEAR in the upstream gradient -- doubling it
    doubles every parameter gradient exactly (a power of two commutes with every rounding of the 16-bit path);
  * fused Adam on the model's 65 M parameters against torch.optim.Adam.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B, T, W, FFT, HPF = 32, 16, 224, 512, 8


def test_extractor_at_the_benched_batch_is_the_per_clip_computation():
    """VideoAttention._inference semantics per clip (video_attention.py:38-103, av_dataset.py:321-333) at 512 frames per call."""
    import maavss_amd
    from oracle import vit_ref_cpu as vref
    sd = vref.seeded_vit_state(3)
    va = maavss_amd.VideoAttention(path_to_weights="/nonexistent.pth")
    va.load_state_dict(sd)
    frames = vref.synthetic_frames(B * T, W, 9)
    full = va.attention_frames(frames.cuda(), clip_frames=T)
    assert full.shape == (B * T, 1, W, W) and bool(torch.isfinite(full).all())
    for c in (0, 13, B - 1):
        alone = va.attention_frames(frames[c * T:(c + 1) * T].cuda(), clip_frames=T)
        assert torch.equal(alone, full[c * T:(c + 1) * T]), f"clip {c} differs inside the batch"
    # every clip is normalised to a maximum of exactly 1 (av_dataset.py:328), every frame is non-negative
    per_clip_max = full.view(B, -1).max(1).values
    assert torch.equal(per_clip_max, torch.ones_like(per_clip_max)) and full.min().item() >= 0.0
    c = 13
    with torch.no_grad():
        ref = vref.clip_normalise_ref(vref.inference_ref(sd, frames[c * T:(c + 1) * T]))          # [1, T, H, W]
    err = (full[c * T:(c + 1) * T].cpu().view(T, W, W) - ref[0]).abs()
    print(f"[parity] full size: clip {c} of the 512-frame batch vs the fp32 oracle: max|err| {err.max().item():.3e} mean {err.mean().item():.3e}")
    assert err.max().item() < 8e-3 and err.mean().item() < 6e-4


def test_stft_at_the_benched_batch():
    """AV_Dataset.stft (av_dataset.py:157-179) on 32 clips of 8448 samples at once."""
    import maavss_amd
    from oracle import stft_ref_cpu as sref
    hop, length, t_a = maavss_amd.calc_hop_size(T, HPF, 30, 16000)
    audio = sref.synthetic_audio(B, length, 5)
    st = maavss_amd.STFT(FFT, hop, noise_std=0.1, device="cuda")
    a = audio.cuda()
    _, y = st(a, want_x=False)
    assert y.shape == (B, 2, t_a, FFT // 2 + 1)
    for i in (0, 17, B - 1):
        _, yi = st(a[i:i + 1], want_x=False)
        assert torch.equal(yi[0], y[i]), f"sample {i} differs inside the batch"
    _, y2 = st(2.0 * a, want_x=False)
    assert torch.equal(y2, 2.0 * y)                                   # exact: scaling by 2 commutes with every f32 rounding
    other = sref.synthetic_audio(B, length, 6).cuda()
    _, yo = st(other, want_x=False)
    _, ys = st(a + other, want_x=False)
    assert (ys - (y + yo)).abs().max().item() < 2e-5                  # additivity to f32 rounding (|y| <= ~0.5)
    ref = sref.stft_ref(audio[[3, 29]], FFT, hop)
    np.testing.assert_allclose(y[[3, 29]].cpu().numpy(), ref.numpy(), rtol=0, atol=5e-6)
    # in-kernel noise at full size: x - y ~ N(0, sigma^2), the same draw for the same seed wherever the sample sits in the batch
    x, _ = st(a, seed=77)
    d = ((x - y) / 0.1).flatten().double()
    assert abs(d.mean().item()) < 2e-3 and abs(d.std().item() - 1) < 2e-3


def _model(precise, seed=43):
    import maavss_amd
    from oracle import avse_ref_cpu as orc
    n_bins, t_a = FFT // 2 + 1, HPF * T
    shapes = ([B, 2, t_a, n_bins], [B, 1, T, W, W], HPF)
    model = maavss_amd.AV_Fusion_Model_Frames(*shapes, precise=precise, spatial_match="adaptive")
    twin = orc.AVFusionFramesRef([2, 2, t_a, n_bins], [2, 1, T, W, W], HPF, spatial_match="adaptive")
    model.load_state_dict(orc.seeded_state_dict(twin, seed), strict=True)
    orc.load_seeded(twin, seed)
    return model.to("cuda"), twin, orc.synthetic_batch(B, T, W, t_a, n_bins, HPF, seed + 1)


@pytest.mark.parametrize("precise", [True, False])
def test_fusion_network_eval_mode_batch_of_32_is_its_pairs(precise):
    """forward (avse_model_final.py:258-274) and backward in eval mode: clips are independent, so the batch of 32 must reproduce its
    16 pairs -- outputs slice by slice, parameter gradients as the sum -- and a pair is small enough for the oracle twin."""
    import maavss_amd
    model, twin, (x_a, x_v, y_a, y_v) = _model(precise)
    model.eval()
    twin.eval()
    x_a, x_v = x_a.cuda(), x_v.cuda()
    g = torch.Generator().manual_seed(5)
    ga = torch.randn(B, 2, HPF, FFT // 2 + 1, generator=g).cuda() / (B * 100)
    gv = torch.randn(B, 1, W, W, generator=g).cuda() / (B * 100)
    prev = maavss_amd.set_deterministic(True)
    try:
        a, v, fused = model(x_a, x_v)
        ((a * ga).sum() + (v * gv).sum()).backward()
        full = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
        a, v, fused = a.detach(), v.detach(), fused.detach()
        acc = {k: torch.zeros_like(t) for k, t in full.items()}
        out_tol = 3e-5 if precise else 2e-4
        for p in range(B // 2):
            for q in model.parameters():
                q.grad = None
            s = slice(2 * p, 2 * p + 2)
            a2, v2, f2 = model(x_a[s], x_v[s])
            assert (a2 - a[s]).abs().max().item() < out_tol and (v2 - v[s]).abs().max().item() < out_tol, p
            assert (f2 - fused[s]).abs().max().item() < out_tol, p
            ((a2 * ga[s]).sum() + (v2 * gv[s]).sum()).backward()
            for k, q in model.named_parameters():
                if q.grad is not None:
                    acc[k] += q.grad
            if p == 7:
                pair_out, pair_grads = (a2.detach().cpu(), v2.detach().cpu()), {k: q.grad.detach().cpu().clone() for k, q in model.named_parameters() if q.grad is not None}
    finally:
        maavss_amd.set_deterministic(prev)
    worst = 0.0
    for k, t in full.items():
        rel = (t - acc[k]).norm().item() / (t.norm().item() + 1e-30)
        worst = max(worst, rel)
        assert rel < (2e-4 if precise else 2e-2), (k, rel)
    # the pair against the oracle twin (fp32, eval mode): outputs and gradients
    s = slice(14, 16)
    a_ref, v_ref, _ = twin(x_a[s].cpu(), x_v[s].cpu())
    ((a_ref * ga[s].cpu()).sum() + (v_ref * gv[s].cpu()).sum()).backward()
    mse = float(((pair_out[0] - a_ref.detach()) ** 2).mean())
    print(f"[parity] full size, eval mode ({'exact f32' if precise else '16-bit'} path): batch of 32 vs the sum of its 16 pairs: worst gradient "
          f"tensor {worst:.2e} relative L2; pair 7 vs the oracle twin: mask-MSE {mse:.3e}")
    assert mse <= (1e-9 if precise else 1e-5)
    ref = dict(twin.named_parameters())
    for k, t in pair_grads.items():
        if k.startswith("stft_autoencoder.") or ref[k].grad is None:
            continue
        rel = (t - ref[k].grad).norm().item() / (ref[k].grad.norm().item() + 1e-30)
        assert rel < (3e-3 if precise else 0.25), (k, rel)


@pytest.mark.parametrize("precise", [True, False])
def test_train_mode_backward_is_linear_in_the_upstream_gradient_at_the_benched_batch(precise):
    """One training forward + backward (train_avse_frames.py:164-170) at B = 32 twice, the second time with the upstream gradients
    doubled: every parameter gradient doubles EXACTLY (deterministic mode: no atomics whose order could differ)."""
    import maavss_amd
    model, _, (x_a, x_v, y_a, y_v) = _model(precise)
    model.train()
    x_a, x_v = x_a.cuda(), x_v.cuda()
    g = torch.Generator().manual_seed(6)
    ga = torch.randn(B, 2, HPF, FFT // 2 + 1, generator=g).cuda() / (B * 100)
    gv = torch.randn(B, 1, W, W, generator=g).cuda() / (B * 100)
    need = {n: True for n, _ in model.named_parameters()}
    prev = maavss_amd.set_deterministic(True)
    try:
        grads = []
        for scale in (1.0, 2.0):
            (a, v, fused), sv = model._engine_forward(x_a, x_v, train=True)
            d_a = (scale * ga).reshape(a.shape).contiguous()
            d_v = (scale * gv).reshape(v.shape).contiguous()
            out = model._engine_backward(sv, d_a, d_v, None, need)
            grads.append({k: t.detach().clone() for k, t in out.items()})
    finally:
        maavss_amd.set_deterministic(prev)
    assert len(grads[0]) >= 30
    for k, t in grads[0].items():
        assert bool(torch.isfinite(t).all()) and t.abs().max().item() > 0, k
        assert torch.equal(grads[1][k], 2.0 * t), (k, (grads[1][k] - 2.0 * t).abs().max().item())


def test_fused_adam_on_the_full_parameter_buffer_matches_torch():
    """torch.optim.Adam (train_avse_frames.py:92,181) on all 65 M parameters of the 224^2 model, three steps, against the one-launch fused kernel."""
    from maavss_amd.trainer import FlatParams, FusedAdam
    model, _, _ = _model(False)
    flat = FlatParams(model)
    opt = FusedAdam(flat, lr=1e-3)
    ref_p = [p.detach().clone().requires_grad_(True) for p in model.parameters()]
    ref_opt = torch.optim.Adam(ref_p, lr=1e-3)
    g = torch.Generator(device="cuda").manual_seed(11)
    for step in range(3):
        flat.grads.copy_(torch.randn(flat.total, device="cuda", generator=g) * 10.0 ** (-step))
        flat.mark(flat.names)
        for p, q in zip(model.parameters(), ref_p):
            q.grad = p.grad.detach().clone()
        opt.step()
        ref_opt.step()
    worst = max((p.detach() - q.detach()).abs().max().item() for p, q in zip(model.parameters(), ref_p))
    assert sum(p.numel() for p in model.parameters()) > 60e6       # 65 M at 224^2 (73 M at 256^2)
    assert worst < 2e-6, worst
