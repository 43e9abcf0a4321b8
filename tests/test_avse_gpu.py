"""GPU parity of the drop-in AV_Fusion_Model_Frames (HIP engine, through the C-ABI) against
(1) the committed golden vectors produced by the reference's own model (oracle/make_golden.py) and
(2) the CPU oracle run live on the same seeded inputs."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _golden(golden_dir, name):
    z = np.load(os.path.join(golden_dir, f"avse_{name}.npz"), allow_pickle=False)
    return z, {k[5:]: z[k].item() for k in z.files if k.startswith("meta_")}


def _build(m, precise, spatial_match="exact"):
    import maavss_amd
    from oracle import avse_ref_cpu as orc
    n_bins, t_a = m["fft_len"] // 2 + 1, m["hops_per_frame"] * m["frames"]
    shapes = ([m["batch"], 2, t_a, n_bins], [m["batch"], 1, m["frames"], m["width"], m["width"]], m["hops_per_frame"])
    model = maavss_amd.AV_Fusion_Model_Frames(*shapes, precise=precise, spatial_match=spatial_match)
    twin = orc.AVFusionFramesRef(*shapes, spatial_match=spatial_match)
    assert list(model.state_dict().keys()) == list(twin.state_dict().keys())
    model.load_state_dict(orc.seeded_state_dict(twin, m["seed"]), strict=True)
    batch = orc.synthetic_batch(m["batch"], m["frames"], m["width"], t_a, n_bins, m["hops_per_frame"], m["seed"] + 1)
    return model.to("cuda"), twin, batch


@pytest.mark.parametrize("name", ["S", "P", "L"])
def test_matches_reference_golden_precise(golden_dir, name):
    """exact-f32 path: outputs, loss, every parameter gradient, BN running stats and one Adam step."""
    import maavss_amd
    z, m = _golden(golden_dir, name)
    model, _, (x_a, x_v, y_a, y_v) = _build(m, precise=True)
    model.train()
    opt = maavss_amd.FusedAdam(model, lr=m["lr"])
    opt.zero_grad()
    a, v, fused = model(x_a.cuda(), x_v.cuda())
    a_loss = torch.nn.functional.mse_loss(a, y_a.cuda())
    v_loss = torch.nn.functional.mse_loss(v, y_v.cuda())
    loss = a_loss + m["loss_coeff"] * v_loss
    loss.backward()
    np.testing.assert_allclose(a.detach().cpu().numpy(), z["x_a_out"], rtol=0, atol=3e-5)
    np.testing.assert_allclose(fused.detach().cpu().numpy(), z["x_av_fused"], rtol=0, atol=3e-5)
    np.testing.assert_allclose(v.detach().flatten()[::997].cpu().numpy(), z["x_v_out_sample"], rtol=0, atol=3e-5)
    mse = float(((a.detach().cpu().numpy() - z["x_a_out"]) ** 2).mean())
    assert mse < 1e-9, mse                       # BASELINE target is 1e-5
    assert abs(loss.item() - z["loss"]) < 2e-6 and abs(a_loss.item() - z["a_loss"]) < 2e-6
    params = dict(model.named_parameters())
    for i, k in enumerate(z["param_names"]):
        k = str(k)
        g = params[k].grad
        if z["grad_norm"][i] < 0:
            assert g is None or float(g.abs().max()) == 0.0, k       # stft_decoder.*: no gradient in the reference
            continue
        gn = g.double().norm().item()
        assert abs(gn - z["grad_norm"][i]) <= 2e-3 * z["grad_norm"][i] + 1e-7, (k, gn, z["grad_norm"][i])
        flat = g.flatten()
        idx = (torch.arange(8) * (flat.numel() - 1)) // 7
        scale = z["grad_norm"][i] / np.sqrt(flat.numel())
        np.testing.assert_allclose(flat[idx.cuda()].cpu().numpy(), z["grad_sample"][i], rtol=5e-3, atol=5e-2 * scale, err_msg=k)
    opt.step()
    for i, k in enumerate(z["param_names"]):
        p = params[str(k)].detach().double()
        # the first Adam step moves every weight by lr*sign(g): a gradient element that is ~0 can flip sign on
        # summation-order noise and shifts the checksum by 2*lr; allow max(2, 0.1 %) such elements per tensor.
        flips = 2 * m["lr"] * max(2, 1e-3 * p.numel())
        assert abs(p.sum().item() - z["adam_wsum"][i]) <= 1e-5 * z["adam_wabs"][i] + 1e-6 + flips, k
        assert abs(p.abs().sum().item() - z["adam_wabs"][i]) <= 1e-5 * z["adam_wabs"][i] + 1e-6 + flips, k
    bufs = dict(model.named_buffers())
    for i, k in enumerate(z["bn_names"]):
        k = str(k)
        assert abs(bufs[k + ".running_mean"].double().sum().item() - z["bn_running_mean_sum"][i]) < 1e-4, k
        assert abs(bufs[k + ".running_var"].double().sum().item() - z["bn_running_var_sum"][i]) < 1e-3, k
        assert bufs[k + ".num_batches_tracked"].item() == 1


@pytest.mark.parametrize("name", ["P"])
def test_16bit_path_within_mask_mse_target(golden_dir, name):
    """16-bit-operand MFMA path (what bench.py runs; f16 forward / bf16 backward operands, f32 accumulate):
    mask-MSE vs the reference <= 1e-5 (BASELINE.json)."""
    z, m = _golden(golden_dir, name)
    model, _, (x_a, x_v, y_a, y_v) = _build(m, precise=False)
    model.train()
    a, v, fused = model(x_a.cuda(), x_v.cuda())
    mse = float(((a.detach().cpu().numpy() - z["x_a_out"]) ** 2).mean())
    assert mse <= 1e-5, mse
    loss = torch.nn.functional.mse_loss(a, y_a.cuda()) + m["loss_coeff"] * torch.nn.functional.mse_loss(v, y_v.cuda())
    assert abs(loss.item() - z["loss"]) < 1e-3
    loss.backward()
    params = dict(model.named_parameters())
    for i, k in enumerate(z["param_names"]):
        if z["grad_norm"][i] > 0:
            gn = params[str(k)].grad.double().norm().item()
            assert abs(gn - z["grad_norm"][i]) <= 0.08 * z["grad_norm"][i] + 1e-7, (str(k), gn, z["grad_norm"][i])


def test_trainstep_equals_autograd_path_and_oracle():
    """The autograd-free TrainStep (bench path) == the nn.Module/autograd path == CPU oracle, incl. Adam."""
    import maavss_amd
    from oracle import avse_ref_cpu as orc
    m = dict(batch=2, frames=8, width=128, fft_len=256, hops_per_frame=8, seed=11)
    model, twin, (x_a, x_v, y_a, y_v) = _build(m, precise=True)
    orc.load_seeded(twin, m["seed"])
    twin.train()
    opt_ref = torch.optim.Adam(twin.parameters(), lr=1e-3)
    step = maavss_amd.TrainStep(model, lr=1e-3, loss_coeff=0.001, num_seq=1)
    for it in range(2):
        opt_ref.zero_grad()
        loss_ref, a_ref, v_ref, _ = orc.loss_ref(twin, x_a, x_v, y_a, y_v, 0.001, 1)
        loss_ref.backward()
        opt_ref.step()
        losses = step(x_a.cuda(), x_v.cuda(), y_a.cuda(), y_v.cuda())
        np.testing.assert_allclose(losses.cpu().numpy(), [a_ref.item(), v_ref.item(), loss_ref.item()], rtol=2e-4, atol=1e-6)
    ref_params = dict(twin.named_parameters())
    for k, p in model.named_parameters():
        if k.startswith("stft_autoencoder."):
            continue
        # Adam's first steps move every weight by ~lr * sign(g): elements whose gradient is ~0 may flip sign on
        # rounding noise, so allow a 1e-4 fraction of outliers (bounded by 2 steps * 2 * lr) and require the rest tight.
        diff = (p.detach().cpu() - ref_params[k].detach()).abs()
        assert diff.max().item() <= 4.1e-3, k
        assert (diff > 5e-5).float().mean().item() <= 1e-3, (k, diff.max().item())


def test_sliding_window_step_matches_reference_loop():
    """TrainStep.sliding_window_step == the reference's num_seq-window optimizer step (train_avse_frames.py:143-181):
    loss / num_seq per window, gradients accumulated, one Adam step -- run on the oracle twin with torch.optim.Adam."""
    import maavss_amd
    from oracle import avse_ref_cpu as orc
    num_seq, num_frames, hpf = 3, 8, 8
    m = dict(batch=2, frames=num_frames, width=128, fft_len=256, hops_per_frame=hpf, seed=17)
    model, twin, _ = _build(m, precise=True)
    orc.load_seeded(twin, m["seed"])
    twin.train()
    n_bins, t_tot = 129, num_frames + num_seq - 1
    g = torch.Generator().manual_seed(23)
    x_stft = torch.rand(2, 2, hpf * t_tot, n_bins, generator=g)
    y_stft = torch.rand(2, 2, hpf * t_tot, n_bins, generator=g) * 0.8
    x_attn = torch.rand(2, 1, t_tot, 128, 128, generator=g)
    y_attn = torch.rand(2, 1, t_tot, 128, 128, generator=g)
    opt = torch.optim.Adam(twin.parameters(), lr=1e-3)
    opt.zero_grad()
    mid = (num_seq - 1) // 2
    outs = []
    for j in range(num_seq):
        yh_a, yh_v, _ = twin(x_stft[:, :, hpf * j:hpf * (j + num_frames)], x_attn[:, :, j:j + num_frames])
        a_loss = torch.nn.functional.mse_loss(yh_a, y_stft[:, :, hpf * (j + mid):hpf * (j + mid + 1)])
        v_loss = torch.nn.functional.mse_loss(yh_v, y_attn[:, :, j + mid])
        loss = (a_loss + 0.001 * v_loss) / num_seq
        loss.backward()
        outs.append(yh_a.detach())
    opt.step()
    step = maavss_amd.TrainStep(model, lr=1e-3, loss_coeff=0.001, num_seq=num_seq)
    losses, out_stft, out_attn = step.sliding_window_step(x_stft.cuda(), y_stft.cuda(), x_attn.cuda(), y_attn.cuda(), num_frames, hpf,
                                                          collect=True)
    np.testing.assert_allclose(losses.cpu().numpy(), [a_loss.item(), v_loss.item(), loss.item()], rtol=2e-4, atol=1e-6)
    assert tuple(out_stft.shape) == (2, 2, hpf * num_seq, n_bins) and tuple(out_attn.shape) == (2, 1, num_seq, 128, 128)
    # windows 1.. run on BatchNorm buffers the earlier windows updated, exactly as in the reference loop
    np.testing.assert_allclose(out_stft.cpu().numpy(), torch.cat(outs, dim=2).numpy(), rtol=0, atol=5e-5)
    ref_params = dict(twin.named_parameters())
    for k, p in model.named_parameters():
        if k.startswith("stft_autoencoder.") or k.startswith("stft_decoder."):
            continue
        diff = (p.detach().cpu() - ref_params[k].detach()).abs()
        assert diff.max().item() <= 2.1e-3, k                      # one Adam step: |dw| <= lr, sign flips of ~0 gradients 2*lr
        # three accumulated windows: a few more gradient elements sit at ~0 than after a single window
        assert (diff > 5e-5).float().mean().item() <= 2e-3, (k, diff.max().item())
    bufs = dict(twin.named_buffers())
    for k, bbuf in model.named_buffers():
        if k.startswith("stft_decoder.") or k.startswith("stft_autoencoder."):
            continue
        if k.endswith("num_batches_tracked"):
            assert bbuf.item() == num_seq
        else:
            np.testing.assert_allclose(bbuf.cpu().numpy(), bufs[k].numpy(), rtol=1e-4, atol=1e-5, err_msg=k)


@pytest.mark.parametrize("batch,frames", [(2, 4), (1, 8)])      # (1, 8): BASELINE config[0], one 8-frame 224^2 clip
def test_adaptive_extension_224_matches_oracle(batch, frames):
    """224^2 is not constructible in the reference; the 'adaptive' extension is checked against the oracle twin."""
    from oracle import avse_ref_cpu as orc
    m = dict(batch=batch, frames=frames, width=224, fft_len=512, hops_per_frame=8, seed=5)
    model, twin, (x_a, x_v, y_a, y_v) = _build(m, precise=True, spatial_match="adaptive")
    orc.load_seeded(twin, m["seed"])
    twin.train()
    loss_ref, _, _, (a_ref, v_ref, f_ref) = orc.loss_ref(twin, x_a, x_v, y_a, y_v, 0.001, 1)
    loss_ref.backward()
    model.train()
    a, v, fused = model(x_a.cuda(), x_v.cuda())
    loss = torch.nn.functional.mse_loss(a, y_a.cuda()) + 0.001 * torch.nn.functional.mse_loss(v, y_v.cuda())
    loss.backward()
    np.testing.assert_allclose(a.detach().cpu().numpy(), a_ref.detach().numpy(), rtol=0, atol=3e-5)
    ref_params = dict(twin.named_parameters())
    for k, p in model.named_parameters():
        if k.startswith("stft_autoencoder.") or ref_params[k].grad is None:
            continue
        gr = ref_params[k].grad
        assert (p.grad.cpu() - gr).norm().item() <= 3e-3 * gr.norm().item() + 1e-7, k


def test_config3_adaptive_384_32_frames_1024pt_matches_oracle():
    """BASELINE config[3]: 32-frame 384^2 clips + 1024-pt STFT (F = 513, T_a = 256).  Like 224^2 this frame size is not
    constructible in the reference (SURVEY finding 2): the 'adaptive' extension against the oracle twin, B = 1, exact-f32."""
    from oracle import avse_ref_cpu as orc
    m = dict(batch=1, frames=32, width=384, fft_len=1024, hops_per_frame=8, seed=29)
    model, twin, (x_a, x_v, y_a, y_v) = _build(m, precise=True, spatial_match="adaptive")
    assert model.s_v == 36 and model.n_bins == 513 and model.t_a == 256
    orc.load_seeded(twin, m["seed"])
    twin.train()
    loss_ref, _, _, (a_ref, v_ref, f_ref) = orc.loss_ref(twin, x_a, x_v, y_a, y_v, 0.001, 1)
    loss_ref.backward()
    model.train()
    a, v, fused = model(x_a.cuda(), x_v.cuda())
    loss = torch.nn.functional.mse_loss(a, y_a.cuda()) + 0.001 * torch.nn.functional.mse_loss(v, y_v.cuda())
    loss.backward()
    np.testing.assert_allclose(a.detach().cpu().numpy(), a_ref.detach().numpy(), rtol=0, atol=5e-5)
    assert abs(loss.item() - loss_ref.item()) < 5e-6
    ref_params = dict(twin.named_parameters())
    for k, p in model.named_parameters():
        if k.startswith("stft_autoencoder.") or ref_params[k].grad is None:
            continue
        gr = ref_params[k].grad
        assert (p.grad.cpu() - gr).norm().item() <= 5e-3 * gr.norm().item() + 1e-7, k


def test_eval_mode_uses_running_statistics():
    """model.eval(): BatchNorm with running statistics, buffers untouched, matches the oracle twin in eval()."""
    from oracle import avse_ref_cpu as orc
    m = dict(batch=2, frames=8, width=128, fft_len=256, hops_per_frame=8, seed=13)
    model, twin, (x_a, x_v, _, _) = _build(m, precise=True)
    orc.load_seeded(twin, m["seed"])
    twin.eval()
    model.eval()
    before = {k: v.clone() for k, v in model.named_buffers()}
    with torch.no_grad():
        a_ref, v_ref, f_ref = twin(x_a, x_v)
        a, v, f = model(x_a.cuda(), x_v.cuda())
    np.testing.assert_allclose(a.cpu().numpy(), a_ref.numpy(), rtol=0, atol=3e-5)
    np.testing.assert_allclose(f.cpu().numpy(), f_ref.numpy(), rtol=0, atol=3e-5)
    np.testing.assert_allclose(v.cpu().numpy(), v_ref.numpy(), rtol=0, atol=3e-5)
    for k, b in model.named_buffers():
        assert torch.equal(b, before[k]), k


@pytest.mark.parametrize("precise", [True, False])
def test_backward_through_an_eval_mode_forward(precise):
    """torch autograd allows loss.backward() after model.eval(): BatchNorm normalises with its running statistics and its
    backward has no batch-mean terms (VERDICT r2 'missing' item 6).  forward() and audio_ae_forward() against the twin in
    eval(); the 16-bit conv modes against the same fp32 twin with the forward-rounding tolerance."""
    from oracle import avse_ref_cpu as orc
    m = dict(batch=2, frames=8, width=128, fft_len=256, hops_per_frame=8, seed=17)
    model, twin, (x_a, x_v, y_a, y_v) = _build(m, precise=precise)
    orc.load_seeded(twin, m["seed"])
    twin.eval()
    model.eval()
    loss_ref, _, _, (a_ref, _, _) = orc.loss_ref(twin, x_a, x_v, y_a, y_v, 0.001, 1)
    loss_ref.backward()
    a, v, _ = model(x_a.cuda(), x_v.cuda())
    loss = torch.nn.functional.mse_loss(a, y_a.cuda()) + 0.001 * torch.nn.functional.mse_loss(v, y_v.cuda())
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < (5e-6 if precise else 2e-4)
    ref_params = dict(twin.named_parameters())
    tol = 3e-3 if precise else 0.25
    for k, p in model.named_parameters():
        if k.startswith("stft_autoencoder.") or ref_params[k].grad is None:
            continue
        gr = ref_params[k].grad
        assert (p.grad.cpu() - gr).norm().item() <= tol * gr.norm().item() + 1e-7, k
    if precise:
        for p in list(model.parameters()) + list(twin.parameters()):
            p.grad = None
        ref = twin.audio_ae_forward(x_a)
        torch.nn.functional.mse_loss(ref, x_a).backward()
        got = model.audio_ae_forward(x_a.cuda())
        torch.nn.functional.mse_loss(got, x_a.cuda()).backward()
        np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().numpy(), rtol=0, atol=3e-5)
        for k, p in model.named_parameters():
            gr = ref_params[k].grad
            if k.startswith("stft_autoencoder.") or gr is None:
                continue
            assert (p.grad.cpu() - gr).norm().item() <= 3e-3 * gr.norm().item() + 1e-7, k


@pytest.mark.parametrize("name", ["S", "P"])
def test_audio_autoencoder_matches_reference_golden(golden_dir, name):
    """audio_ae_forward through the HIP engine (conv2d / convt2d / BN kernels) vs the reference's own numbers:
    output, loss, every encoder and decoder gradient, BN running statistics (train_audio_net.py:107-109)."""
    from oracle import avse_ref_cpu as orc
    z = np.load(os.path.join(golden_dir, f"avse_ae_{name}.npz"), allow_pickle=False)
    m = {k[5:]: z[k].item() for k in z.files if k.startswith("meta_")}
    model, _, (x_a, _, _, _) = _build(m, precise=True)
    model.train()
    y = x_a.cuda()
    yh = model.audio_ae_forward(y)
    assert yh.shape == y.shape
    loss = torch.nn.functional.mse_loss(yh, y)
    loss.backward()
    np.testing.assert_allclose(yh.detach().flatten()[::61].cpu().numpy(), z["yh_sample"], rtol=0, atol=2e-5)
    assert abs(yh.detach().double().sum().item() - z["yh_sum"]) <= 1e-5 * z["yh_abs_sum"] + 1e-4
    assert abs(loss.item() - z["loss"]) < 2e-6
    params = dict(model.named_parameters())
    for i, k in enumerate(z["param_names"]):
        g = params[str(k)].grad
        gn = g.double().norm().item()
        assert abs(gn - z["grad_norm"][i]) <= 2e-3 * z["grad_norm"][i] + 1e-7, (str(k), gn, z["grad_norm"][i])
        flat = g.flatten()
        idx = (torch.arange(8) * (flat.numel() - 1)) // 7
        scale = z["grad_norm"][i] / np.sqrt(flat.numel())
        np.testing.assert_allclose(flat[idx.cuda()].cpu().numpy(), z["grad_sample"][i], rtol=5e-3, atol=5e-2 * scale, err_msg=str(k))
    bufs = dict(model.named_buffers())
    for i, k in enumerate(z["bn_names"]):
        k = str(k)
        assert abs(bufs[k + ".running_mean"].double().sum().item() - z["bn_running_mean_sum"][i]) < 1e-4, k
        assert abs(bufs[k + ".running_var"].double().sum().item() - z["bn_running_var_sum"][i]) < 1e-3, k
    # forward() gives the decoder no gradient (reference behaviour) and still works after an autoencoder step
    assert all(params[n].grad is not None for n in params if n.startswith("stft_decoder."))


def test_audio_autoencoder_eval_mode_matches_oracle():
    from oracle import avse_ref_cpu as orc
    m = dict(batch=2, frames=8, width=128, fft_len=256, hops_per_frame=8, seed=21)
    model, twin, (x_a, _, _, _) = _build(m, precise=True)
    orc.load_seeded(twin, m["seed"])
    twin.eval()
    model.eval()
    with torch.no_grad():
        ref = twin.audio_ae_forward(x_a)
        got = model.audio_ae_forward(x_a.cuda())
    np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=0, atol=3e-5)


def test_reference_constructor_guards():
    import maavss_amd
    with pytest.raises(ValueError):
        maavss_amd.AV_Fusion_Model_Frames([1, 2, 128, 257], [1, 1, 16, 224, 224], 8)
    with pytest.raises(ValueError):
        maavss_amd.AV_Fusion_Model_Frames([1, 2, 64, 257], [1, 1, 8, 256, 256], 8, latent_channels=64)
    model = maavss_amd.AV_Fusion_Model_Frames([1, 2, 64, 257], [1, 1, 8, 256, 256], 8)
    with pytest.raises(maavss_amd._lib.MaavssError):
        model(torch.zeros(1, 2, 64, 257), torch.zeros(1, 1, 8, 256, 256))      # CPU tensors: no fallback


def test_deterministic_mode_is_bit_reproducible():
    """maavss_amd.set_deterministic(True): the split-K forms of the M = batch Linear layers accumulate with f32
    atomics by default (reproducible to summation order); with the switch set, two runs of three optimizer steps from the same state
    end bit-identical (the conv path, BatchNorm and Adam are deterministic already), and the result stays within summation-order
    distance of the default mode's."""
    import maavss_amd
    m = dict(batch=2, frames=8, width=128, fft_len=256, hops_per_frame=8, seed=23)

    def run():
        model, _, (x_a, x_v, y_a, y_v) = _build(m, precise=False)
        model.train()
        step = maavss_amd.TrainStep(model, lr=1e-4)
        losses = [step(x_a.cuda(), x_v.cuda(), y_a.cuda(), y_v.cuda())[2].item() for _ in range(3)]
        torch.cuda.synchronize()
        return losses, step.flat.params.clone()

    prev = maavss_amd.set_deterministic(True)
    try:
        l1, p1 = run()
        l2, p2 = run()
    finally:
        maavss_amd.set_deterministic(prev)
    assert l1 == l2 and torch.equal(p1, p2)
    l3, p3 = run()                                       # default (atomic split-K) mode
    assert abs(l3[0] - l1[0]) <= 1e-6 * abs(l1[0]) + 1e-8
    assert (p3 - p1).norm().item() <= 1e-4 * p1.norm().item()


@pytest.mark.parametrize("degenerate", [False, True])
def test_first_layer_without_its_conv_output_equals_the_storing_path_in_the_model(degenerate):
    """The 16-bit training path never stores the first layer's conv output (three recompute passes); `c1_recompute=False`
    selects the kernels that store and re-read it.  Same weights, same batch: outputs, every gradient and the BatchNorm running statistics
    are BIT-IDENTICAL (deterministic mode).  degenerate: one first-layer BatchNorm weight below 1e-2 -- x-hat is then not recoverable from
    the pooled output, the statistics pass stores y after all and the backward reduction gathers from it."""
    import maavss_amd
    m = dict(batch=2, frames=8, width=128, fft_len=256, hops_per_frame=8, seed=31)
    prev = maavss_amd.set_deterministic(True)
    try:
        res = []
        for recompute in (True, False):
            model, _, (x_a, x_v, y_a, y_v) = _build(m, precise=False)
            model.c1_recompute = recompute
            if degenerate:
                with torch.no_grad():
                    model.visual_encoder[1].weight[5] = 2e-3
            model.train()
            a, v, f = model(x_a.cuda(), x_v.cuda())
            loss = torch.nn.functional.mse_loss(a, y_a.cuda()) + 0.001 * torch.nn.functional.mse_loss(v, y_v.cuda())
            loss.backward()
            res.append((a.detach(), v.detach(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None},
                        {k: b.detach().clone() for k, b in model.named_buffers()}))
    finally:
        maavss_amd.set_deterministic(prev)
    (a0, v0, g0, b0), (a1, v1, g1, b1) = res
    assert torch.equal(a0, a1) and torch.equal(v0, v1)
    assert g0.keys() == g1.keys() and len(g0) > 30
    for k in g0:
        assert torch.equal(g0[k], g1[k]), (k, (g0[k] - g1[k]).abs().max().item())
    for k in b0:
        assert torch.equal(b0[k], b1[k]), k
    assert bool(torch.isfinite(g0["visual_encoder.1.weight"]).all()) and g0["visual_encoder.1.weight"].abs().max().item() > 0
