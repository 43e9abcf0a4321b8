"""Round-2 parity gates ("Next round" items 1 and 8), all through the C-ABI:

  * the configuration bench.py actually runs -- T = 16, 224^2, spatial_match="adaptive", default 16-bit conv modes --
    against the fp32 oracle twin (mask-MSE, loss) and against a twin that rounds the conv operands where the kernels do
    (per-parameter gradient SAMPLES, tight);
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

# The 16-bit path rounds conv operands (forward: IEEE half, backward: bf16).  What that does to the GRADIENTS was measured on the
# CPU oracle one rounding at a time (tests/tools/grad_rounding_ablation.py -> profiles/r3_grad_rounding_ablation.txt):
#   * the bf16 backward operands move the gradient tensors by 0.3-0.6 % relative L2 (1 % on one BatchNorm bias at the benched shape);
#   * the IEEE-half FORWARD operands move them by 7-8 % on the first conv layers (8-11 % on visual_encoder.1.bias; the HIP path
#     logged 8-17 % in round 2), bf16 forward operands by 28 %, a 16-bit mantissa by 1 %: error ~ sqrt(perturbation).  The encoder
#     gradients of these seeded-random problems are sums of ~1e6 cancelling contributions routed by MaxPool argmax / LeakyReLU
#     sign; a forward perturbation eps re-routes a fraction ~eps of them.  It is a property of the test problem's conditioning
#     (the exact-f32 HIP path, 3e-6 forward distance, already sits 2-3e-3 from the oracle), not of a kernel, and splitting dy
#     into bf16 hi + lo in the first layers' weight gradients changes nothing: 7.22 % -> 7.23 %.
# Two computations that round at the same points also decorrelate layer by layer (a value next to a rounding boundary flips;
# tests/tools/fwd_stage_diff.py: 2e-7 after conv0, 3e-4 after conv4), so the emulating twin is matched within a fraction of that
# envelope, not element-exactly.  The gates below therefore are: (1) an ABSOLUTE cap on every gradient tensor against the fp32
# oracle at ~2x the measured forward-rounding floor, plus direction (cosine); (2) not farther from the emulating
# twin than that twin is from fp32; (3) the forward outputs, absolutely (mask-MSE 1e-5, loss).  What training sees over several
# steps is gated by tests/test_parity_r3_gpu.py (10-step trajectory against the fp32 twin).
EMU_SAMPLE_TOL = 0.12      # |g - g_emulated| <= tol * (|g_emulated| + rms(g_emulated)) for every sampled element
EMU_L2_TOL = 0.05          # relative L2 distance of each gradient tensor to the emulating oracle
FP32_L2_CAP = 0.25         # relative L2 distance of ANY gradient tensor to the fp32 oracle (forward-rounding floor: 0.08-0.12 on the CPU)
FP32_COS_MIN = 0.97        # and its direction


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


def _grad_report(model, twin_emu, twin_f32, tag):
    emu, f32 = dict(twin_emu.named_parameters()), dict(twin_f32.named_parameters())
    worst_s, worst_l2, worst_q, worst_cos, worst_ratio = 0.0, 0.0, 0.0, 1.0, 0.0
    for k, p in model.named_parameters():
        if k.startswith("stft_autoencoder.") or f32[k].grad is None:
            continue
        g, ge, gf = p.grad.detach().cpu().flatten(), emu[k].grad.flatten(), f32[k].grad.flatten()
        rms = ge.norm().item() / np.sqrt(ge.numel())
        l2 = (g - ge).norm().item() / (ge.norm().item() + 1e-30)
        quant = (g - gf).norm().item() / (gf.norm().item() + 1e-30)
        idx = torch.randperm(ge.numel(), generator=torch.Generator().manual_seed(7))[:256]
        rel = ((g[idx] - ge[idx]).abs() / (ge[idx].abs() + rms)).max().item()
        worst_s, worst_l2, worst_q = max(worst_s, rel), max(worst_l2, l2), max(worst_q, quant)
        env = (ge - gf).norm().item() / (gf.norm().item() + 1e-30)          # the rounding envelope: emulation vs fp32
        if k == "visual_encoder.0.weight":
            print(f"[parity] {tag}: {k}: L2 to the emulating oracle {l2:.3e}, to the fp32 oracle {quant:.3e}; emulation to fp32 (envelope) {env:.3e}")
        # about as far from the twin that rounds at the same points as that twin is from fp32: two realisations of the same re-routing
        # decorrelate (header).  Factors from EIGHT seeded problems (round 2's two, config[3], the benched shape, three more seeds on P in
        # tests/test_parity_r3_gpu.py), each with margin: tensor L2 up to 1.72 env, distance to fp32 up to 2.03 env, the worst of 256 sampled elements up to 6.1 env
        # (heavy-tailed: one re-routed window moves a weight of a late, small layer by several sigma).  Round 2's factors (1.25 / 3.5) were
        # fitted to its two problems and failed on the new seeds; these relative gates only catch a path that leaves the emulation's
        # neighbourhood -- the guarantees are the ABSOLUTE gates below and the trajectory test.
        assert l2 <= max(EMU_L2_TOL, 2.5 * env), (tag, k, "L2 vs emulating oracle", l2, env)
        assert rel <= max(EMU_SAMPLE_TOL, 8.0 * env), (tag, k, "sampled element vs emulating oracle", rel, env)
        # the HIP path and the emulating twin are two realisations of the same re-routing process (same rounding points, different
        # summation order inside the MFMA): their distances to fp32 have the same statistics, not the same value.  Measured ratio
        # quant / env over the builds of rounds 2-3 and eight seeded problems (each new accumulation order of a conv kernel, each seed
        # re-draws it): 0.9 ... 2.03.
        # (a tensor whose emulated envelope happens to be tiny -- the last BatchNorm bias on one seed: 0.07 % -- may still sit 1.2 % from fp32)
        assert quant <= max(2.5 * env + 2e-3, 3e-2), (tag, k, "L2 vs fp32 oracle outside the operand-rounding envelope", quant, env)
        if env > 1e-2:
            worst_ratio = max(worst_ratio, quant / env)
        cos = torch.dot(g.double(), gf.double()).item() / (g.double().norm().item() * gf.double().norm().item() + 1e-300)
        worst_cos = min(worst_cos, cos)
        assert quant <= FP32_L2_CAP and cos >= FP32_COS_MIN, (tag, k, "absolute gate vs the fp32 oracle", quant, cos)
    print(f"[parity] {tag}: gradients vs 16-bit-emulating oracle: worst tensor L2 {worst_l2:.2e}, worst sampled element "
          f"{worst_s:.2e} of (|g|+rms); vs fp32 oracle (= the IEEE-half forward operands re-routing cancelling contributions): worst tensor L2 {worst_q:.2e}, worst cosine {worst_cos:.4f}; "
          f"largest (distance to fp32) / (distance of the emulating twin to fp32) among tensors with an envelope above 1 %: {worst_ratio:.2f}")
    return worst_ratio


# (the pinned shape P runs the same gates in tests/test_parity_r3_gpu.py::test_gradient_envelope_ratio_over_seeds)
@pytest.mark.parametrize("tag,batch,frames,width,spatial", [("benched T=16 224^2 adaptive", 2, 16, 224, "adaptive")])
def test_16bit_configuration_against_emulating_and_fp32_oracles(tag, batch, frames, width, spatial):
    """BASELINE config[1] as bench.py runs it (B reduced to 2 for the CPU oracle): T=16, 224^2, adaptive, 16-bit modes;
    and the same on the reference-pinned shape P."""
    from oracle import avse_ref_cpu as orc
    model, twin, (x_a, x_v, y_a, y_v) = _build(batch, frames, width, 512, 41, precise=False, spatial_match=spatial)
    shapes = (twin.stft_shape, twin.frame_shape, twin.output_stft_frames)
    emu = orc.AVFusionFramesRef(*shapes, spatial_match=spatial, emulate_16bit=True)
    orc.load_seeded(emu, 41)
    emu.train()
    loss_ref, a_loss_ref, v_loss_ref, (a_ref, v_ref, f_ref) = orc.loss_ref(twin, x_a, x_v, y_a, y_v, 0.001, 1)
    loss_ref.backward()
    loss_emu, _, _, (a_emu, v_emu, _) = orc.loss_ref(emu, x_a, x_v, y_a, y_v, 0.001, 1)
    loss_emu.backward()
    a, v, fused = model(x_a.cuda(), x_v.cuda())
    loss = F.mse_loss(a, y_a.cuda()) + 0.001 * F.mse_loss(v, y_v.cuda())
    loss.backward()
    mse = float(((a.detach().cpu() - a_ref.detach()) ** 2).mean())
    mse_emu = float(((a.detach().cpu() - a_emu.detach()) ** 2).mean())
    print(f"[parity] {tag}: mask-MSE vs fp32 oracle {mse:.3e} (vs emulating oracle {mse_emu:.3e}), "
          f"|dloss| {abs(loss.item() - loss_ref.item()):.3e}")
    assert mse <= 1e-5, mse                                             # BASELINE.json: mask MSE within 1e-5
    assert mse_emu <= 1e-7, mse_emu
    assert abs(loss.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item()) + 1e-5
    assert (v.detach().cpu() - v_ref.detach()).abs().max().item() < 2e-3
    _grad_report(model, emu, twin, tag)


FP8_MASK_MSE_BOUND = 1e-2      # stated bound for attn_dtype="fp8" (measured value printed and in DESIGN.md)


@pytest.mark.parametrize("act", ["f16", "fp8", "bf16"])       # "bf16": selectable storage format; 2.3e-4 -- why half is the default (restored to the suite in round 4)
def test_end_to_end_frames_to_mask_with_the_vit_in_the_loop(act):
    """frames -> attention frames (16-bit HIP ViT) -> AVSE (16-bit HIP) vs the all-fp32 oracle chain on the pinned P shape
    (av_dataset.py:321-333 -> train_avse_frames.py:164-168).  With IEEE-half storage in the extractor (the default) the chain
    meets BASELINE's mask-MSE <= 1e-5; with bf16 storage the extractor's quantisation error alone moves the mask by ~2e-4
    (the fusion network amplifies its input perturbation ~6x), which is why half is the default: reported, bounded."""
    import maavss_amd
    from oracle import avse_ref_cpu as orc, vit_ref_cpu as vref
    b, t, w, hpf = 2, 8, 256, 8
    model, twin, (x_a, _, y_a, _) = _build(b, t, w, 512, 43, precise=False, spatial_match="exact")
    sd = vref.seeded_vit_state(3)
    if act == "fp8":     # BASELINE config[4]: block-scaled fp8 Q K^T / P V inside the IEEE-half extractor
        va = maavss_amd.VideoAttention(path_to_weights="/nonexistent.pth", act_dtype="f16", attn_dtype="fp8")
    else:
        va = maavss_amd.VideoAttention(path_to_weights="/nonexistent.pth", act_dtype=act)
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
    print(f"[parity] end to end ({act} ViT): attention maps max|err| {map_err.max().item():.3e} mean {map_err.mean().item():.3e}; "
          f"mask-MSE {mse:.3e}; |dloss| {abs(loss.item() - loss_ref.item()):.3e} (loss {loss_ref.item():.5f})")
    if act == "fp8":
        # e4m3 operands (3 mantissa bits) cannot meet the 1e-5 target: the mode is selectable (config[4]), never the default
        assert mse <= FP8_MASK_MSE_BOUND, mse
        assert abs(loss.item() - loss_ref.item()) <= 5e-3
    elif act == "f16":
        assert mse <= 1e-5, mse                                   # BASELINE.json: mask MSE within 1e-5 of the reference
        assert abs(loss.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item()) + 1e-5      # same bound as the benched-shape test above
    else:
        assert mse <= 1e-3, mse
        assert abs(loss.item() - loss_ref.item()) <= 5e-4


@pytest.mark.parametrize("width,frames,act", [(64, 4, "f16"), (64, 4, "bf16"), (224, 2, "f16")])
def test_video_attention_matches_rounding_emulating_oracle(width, frames, act):
    """Against an oracle that rounds exactly where the kernels store 16-bit values (weights, LayerNorm output, q/k/v, the
    tile-wise P of the flash loop with its deferred maximum, attention output, polynomial-GELU output): what remains is
    summation order and hardware exp2 / rsqrt ulps, far below the quantisation error itself -- a wrong position-embedding
    row, LayerNorm eps or softmax scale cannot hide in it.  The distance to the fp32 oracle (= the quantisation error of
    the storage format) is printed next to it."""
    import maavss_amd
    from oracle import vit_ref_cpu as vref
    sd = vref.seeded_vit_state(3)
    va = maavss_amd.VideoAttention(path_to_weights="/nonexistent.pth", act_dtype=act)
    va.load_state_dict(sd)
    fr = vref.synthetic_frames(frames, width, 5)
    with torch.no_grad():
        want_emu = vref.inference_ref(sd, fr, emulate=act)
        want_f32 = vref.inference_ref(sd, fr)
        cls_emu = vref.cls_attention(sd, fr, emulate=act)
    got = va._inference(fr)
    got_cls = va.cls_attention(fr.cuda()).cpu()
    e_emu, e_f32 = (got - want_emu).abs().max().item(), (got - want_f32).abs().max().item()
    print(f"[parity] ViT {act} {width}^2: maps max|err| vs rounding-emulating oracle {e_emu:.3e} (mean {(got - want_emu).abs().mean().item():.2e}); "
          f"vs fp32 oracle {e_f32:.3e} = quantisation error")
    # IEEE half (default): <= 5e-3 as VERDICT r1 asked; bf16: its rounding noise is 8x larger and so is the floor at which two
    # computations with the same rounding points decorrelate (LayerNorm / softmax amplify single-ulp flips)
    lim_max, lim_mean, lim_cls = (5e-3, 5e-4, 5e-3) if act == "f16" else (4e-2, 5e-3, 4e-2)
    assert e_emu <= lim_max, e_emu
    assert (got - want_emu).abs().mean().item() <= lim_mean
    rel = (got_cls - cls_emu).abs().max().item() / cls_emu.abs().max().item()
    assert rel <= lim_cls, rel


@pytest.mark.parametrize("width,frames,act", [(64, 4, "f16"), (224, 2, "f16"), (64, 4, "bf16")])
def test_video_attention_with_norm1_behind_the_qkv_product_matches_its_emulation(width, frames, act):
    """VideoAttention(qkv_ln="post") (round 4): the attn.qkv GEMM runs on the rounded RAW rows with gamma-folded weights and applies the row
    statistics in its epilogue.  Against the oracle that rounds exactly there (emulate="<act>-lnpost"), same limits as the default mode; its
    distance to the fp32 oracle is printed next to the default mode's."""
    import maavss_amd
    from oracle import vit_ref_cpu as vref
    sd = vref.seeded_vit_state(3)
    va = maavss_amd.VideoAttention(path_to_weights="/nonexistent.pth", act_dtype=act, qkv_ln="post")
    va.load_state_dict(sd)
    va0 = maavss_amd.VideoAttention(path_to_weights="/nonexistent.pth", act_dtype=act, qkv_ln="pre")
    va0.load_state_dict(sd)
    fr = vref.synthetic_frames(frames, width, 5)
    with torch.no_grad():
        want_emu = vref.inference_ref(sd, fr, emulate=act + "-lnpost")
        want_f32 = vref.inference_ref(sd, fr)
    got, got0 = va._inference(fr), va0._inference(fr)
    e_emu, e_f32, e0_f32 = (got - want_emu).abs().max().item(), (got - want_f32).abs().max().item(), (got0 - want_f32).abs().max().item()
    print(f"[parity] ViT {act} {width}^2 qkv_ln=post: maps max|err| vs its rounding-emulating oracle {e_emu:.3e} (mean {(got - want_emu).abs().mean().item():.2e}); "
          f"vs fp32 oracle {e_f32:.3e} mean {(got - want_f32).abs().mean().item():.2e} (qkv_ln=pre: {e0_f32:.3e} mean {(got0 - want_f32).abs().mean().item():.2e})")
    lim_max, lim_mean = (5e-3, 5e-4) if act == "f16" else (4e-2, 5e-3)
    assert e_emu <= lim_max, e_emu
    assert (got - want_emu).abs().mean().item() <= lim_mean


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
