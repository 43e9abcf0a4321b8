"""Round-3 parity gates ("Next round" item 1), all through the C-ABI:

  * BASELINE config[3] in the mode its bench line runs: T = 32, 384^2, 1024-pt STFT, spatial_match="adaptive", 16-bit conv
    modes (precise=False), B = 1, against the fp32 oracle twin (mask-MSE <= 1e-5, loss) and the rounding-emulating twin;
  * a TRAJECTORY: 10 optimizer steps of the default 16-bit TrainStep against 10 steps of the fp32 oracle twin under
    torch.optim.Adam on the pinned shape P (train_avse_frames.py:150-181: same loss, same Adam), per-step loss within
    1e-3 relative, final-weight drift reported and bounded.

Why the trajectory and not a tight per-tensor gradient bound is the gate for the 16-bit path: on these seeded-random inputs the
encoder gradients are sums of ~1e6 cancelling contributions routed by MaxPool argmax / LeakyReLU sign; ANY forward perturbation
eps re-routes a fraction ~eps of them, so the relative L2 distance of a gradient tensor scales as sqrt(eps) -- measured on the
CPU oracle (tests/tools/grad_rounding_ablation.py -> profiles/r3_grad_rounding_ablation.txt): IEEE-half forward operands (11 bits)
7 %, 16 mantissa bits 1.1 %, and the bf16 backward operands the round-2 verdict suspected 0.4-0.5 % (dy as bf16 hi + lo in the
first two weight gradients: 7.22 % -> 7.23 %, nothing).  What training sees is the trajectory below.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_parity_r2_gpu import _build, _grad_report

pytestmark = pytest.mark.gpu


def test_config3_16bit_modes_against_fp32_and_emulating_oracles():
    """BASELINE config[3] as `bench.py --frames 32 --framesize 384 --fft_len 1024` runs it (B reduced to 1 for the CPU oracle)."""
    from oracle import avse_ref_cpu as orc
    tag = "config[3] T=32 384^2 fft 1024 adaptive"
    model, twin, (x_a, x_v, y_a, y_v) = _build(1, 32, 384, 1024, 29, precise=False, spatial_match="adaptive")
    assert model.s_v == 36 and model.n_bins == 513 and model.t_a == 256
    shapes = (twin.stft_shape, twin.frame_shape, twin.output_stft_frames)
    emu = orc.AVFusionFramesRef(*shapes, spatial_match="adaptive", emulate_16bit=True)
    orc.load_seeded(emu, 29)
    emu.train()
    loss_ref, _, _, (a_ref, v_ref, _) = orc.loss_ref(twin, x_a, x_v, y_a, y_v, 0.001, 1)
    loss_ref.backward()
    loss_emu, _, _, (a_emu, _, _) = orc.loss_ref(emu, x_a, x_v, y_a, y_v, 0.001, 1)
    loss_emu.backward()
    a, v, fused = model(x_a.cuda(), x_v.cuda())
    loss = F.mse_loss(a, y_a.cuda()) + 0.001 * F.mse_loss(v, y_v.cuda())
    loss.backward()
    mse = float(((a.detach().cpu() - a_ref.detach()) ** 2).mean())
    mse_emu = float(((a.detach().cpu() - a_emu.detach()) ** 2).mean())
    print(f"[parity] {tag}: mask-MSE vs fp32 oracle {mse:.3e} (vs emulating oracle {mse_emu:.3e}), "
          f"|dloss| {abs(loss.item() - loss_ref.item()):.3e} (loss {loss_ref.item():.5f})")
    assert mse <= 1e-5, mse                                             # BASELINE.json: mask MSE within 1e-5
    assert mse_emu <= 1e-6, mse_emu
    assert abs(loss.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item()) + 1e-5
    assert (v.detach().cpu() - v_ref.detach()).abs().max().item() < 2e-3
    _grad_report(model, emu, twin, tag)


TRAJ_STEPS = 10


# The suite's wall time is CPU-oracle time (10 twin steps = 16 s per case), so only the gate VERDICT asked for runs here -- the reference's
# lr = 1e-5.  Measured with the same function and recorded in profiles/r3_trajectory_seeds.log: seed 7 (1.6e-4), and at
# lr = 1e-4 the 16-bit path (1.7e-3, gate 5e-3) next to the exact-f32 HIP path as control (1.1e-4): (1e-4, False, 5e-3, 53), (1e-4, True, 5e-3, 53).
@pytest.mark.parametrize("lr,precise,loss_tol,seed", [
    (1e-5, False, 1e-3, 53),
    pytest.param(1e-5, False, 1e-3, 19, marks=pytest.mark.slow),          # 2.0e-4 (profiles/r3_trajectory_seeds.log)
    pytest.param(1e-4, False, 5e-3, 53, marks=pytest.mark.slow),          # 1.7e-3
    pytest.param(1e-4, True, 5e-3, 53, marks=pytest.mark.slow),           # the exact-f32 control: 1.1e-4
])
def test_ten_step_trajectory_of_the_16bit_path_follows_the_fp32_twin(lr, precise, loss_tol, seed):
    """10 x (forward, loss, backward, Adam) on the pinned shape P, the same batch every step (train_avse_frames.py:150-181 with
    num_seq = 1): default 16-bit HIP TrainStep vs the fp32 oracle twin + torch.optim.Adam.  lr = 1e-5 is the reference's
    default (run_config.py:7) and carries VERDICT's 1e-3 gate; at 1e-4 the weights move ten times further per step and the two
    trajectories separate faster -- the exact-f32 HIP path runs as the control for how much of that is the 16-bit path."""
    import maavss_amd
    from oracle import avse_ref_cpu as orc
    prev = maavss_amd.set_deterministic(True)       # reproducible figures: no atomic summation order in the Linear kernels
    try:
        _trajectory(maavss_amd, orc, lr, precise, loss_tol, seed)
    finally:
        maavss_amd.set_deterministic(prev)


def _trajectory(maavss_amd, orc, lr, precise, loss_tol, seed):
    model, twin, (x_a, x_v, y_a, y_v) = _build(2, 8, 256, 512, seed, precise=precise, spatial_match="exact")
    w0 = {k: p.detach().clone() for k, p in twin.named_parameters()}
    opt = torch.optim.Adam(twin.parameters(), lr=lr)
    step = maavss_amd.TrainStep(model, lr=lr, loss_coeff=0.001, num_seq=1)
    xa, xv, ya, yv = x_a.cuda(), x_v.cuda(), y_a.cuda(), y_v.cuda()
    ref_losses, got_losses = [], []
    for _ in range(TRAJ_STEPS):
        opt.zero_grad()
        loss_ref, *_ = orc.loss_ref(twin, x_a, x_v, y_a, y_v, 0.001, 1)
        loss_ref.backward()
        opt.step()
        ref_losses.append(loss_ref.item())
        got_losses.append(step(xa, xv, ya, yv)[2].item())
    rel = [abs(g - r) / abs(r) for g, r in zip(got_losses, ref_losses)]
    # drift of the weights: |w_hip - w_twin| against how far the twin itself moved, per tensor
    worst, worst_k, tot_d, tot_m = 0.0, None, 0.0, 0.0
    for k, p in twin.named_parameters():
        if p.grad is None:
            continue
        moved = (p.detach() - w0[k]).double().norm().item()
        drift = (step.flat.param_views[k].cpu() - p.detach()).double().norm().item()
        tot_d, tot_m = tot_d + drift ** 2, tot_m + moved ** 2
        if moved > 0 and drift / moved > worst:
            worst, worst_k = drift / moved, k
    total = (tot_d / tot_m) ** 0.5
    print(f"[trajectory] seed {seed} lr {lr:g} {'exact-f32' if precise else '16-bit'} HIP path: loss fp32 twin {ref_losses[0]:.6f} -> {ref_losses[-1]:.6f}, HIP {got_losses[0]:.6f} -> "
          f"{got_losses[-1]:.6f}; per-step |dloss|/loss max {max(rel):.2e} (step {rel.index(max(rel))}); weight drift / distance moved: "
          f"all parameters {total:.3e}, worst tensor {worst_k} {worst:.3e}")
    assert ref_losses[-1] < ref_losses[0]                                  # the twin is training
    assert max(rel) <= loss_tol, rel                                       # 1e-3 at the reference's lr
    # after 10 Adam steps the 16-bit path has moved the weights to within this fraction of where fp32 moved them
    assert total <= 0.15, total
    assert worst <= 0.5, (worst_k, worst)


@pytest.mark.parametrize("tag,b,t,w,spatial,model_seed,vit_seed,frame_seed", [
    ("P shape, second seed set", 2, 8, 256, "exact", 101, 11, 21),
    ("benched shape T=16 224^2 adaptive", 1, 16, 224, "adaptive", 43, 3, 9),
    pytest.param("P shape, third seed set", 2, 8, 256, "exact", 57, 5, 33, marks=pytest.mark.slow),
    pytest.param("benched shape, second seed set", 1, 16, 224, "adaptive", 71, 13, 17, marks=pytest.mark.slow),
])      # the two larger of four measured cases (profiles/r3_e2e_seeds.log: 5.4e-6, 3.2e-6, 6.5e-6, 3.0e-6): CPU-oracle time
def test_end_to_end_mask_mse_over_seeds_and_at_the_benched_shape(tag, b, t, w, spatial, model_seed, vit_seed, frame_seed):
    """the end-to-end gate (frames -> IEEE-half HIP ViT -> 16-bit HIP fusion network vs the all-fp32 oracle
    chain, av_dataset.py:321-333 -> train_avse_frames.py:164-168) held on one seed set and one shape.  Same gate, other weights /
    frames / audio, and the shape bench.py runs (T = 16, 224^2, adaptive spatial match)."""
    import maavss_amd
    from oracle import avse_ref_cpu as orc, vit_ref_cpu as vref
    model, twin, (x_a, _, y_a, _) = _build(b, t, w, 512, model_seed, precise=False, spatial_match=spatial)
    sd = vref.seeded_vit_state(vit_seed)
    va = maavss_amd.VideoAttention(path_to_weights="/nonexistent.pth")            # defaults: IEEE-half storage, f16 attention
    va.load_state_dict(sd)
    frames = vref.synthetic_frames(b * t, w, frame_seed)
    with torch.no_grad():
        x_v_ref = torch.stack([vref.clip_normalise_ref(vref.inference_ref(sd, frames[i * t:(i + 1) * t])) for i in range(b)])
    x_v = va.attention_frames(frames.cuda(), clip_frames=t).view(b, 1, t, w, w)
    y_v_ref, y_v = x_v_ref[:, :, t // 2], x_v[:, :, t // 2]
    loss_ref, _, _, (a_ref, _, _) = orc.loss_ref(twin, x_a, x_v_ref, y_a, y_v_ref, 0.001, 1)
    a, v, _ = model(x_a.cuda(), x_v)
    loss = F.mse_loss(a, y_a.cuda()) + 0.001 * F.mse_loss(v, y_v)
    map_err = (x_v.cpu() - x_v_ref).abs()
    mse = float(((a.detach().cpu() - a_ref.detach()) ** 2).mean())
    print(f"[parity] end to end, {tag}: attention maps max|err| {map_err.max().item():.3e} mean {map_err.mean().item():.3e}; "
          f"mask-MSE {mse:.3e}; |dloss| {abs(loss.item() - loss_ref.item()):.3e} (loss {loss_ref.item():.5f})")
    assert mse <= 1e-5, mse                                                       # BASELINE.json: mask MSE within 1e-5 of the reference
    # the loss is a mean of (mask - target)^2: a mask error of mean square m moves it by at most 2 sqrt(loss m) (3.7e-3 at m = 1e-5);
    # measured 2e-6 ... 8e-5 (2e-4 relative) over these cases, gated at 5e-4 relative
    assert abs(loss.item() - loss_ref.item()) <= 5e-4 * abs(loss_ref.item())


@pytest.mark.parametrize("seed", [53, pytest.param(7, marks=pytest.mark.slow), pytest.param(19, marks=pytest.mark.slow)])           # the worst of three (ratio 2.03); seeds 7 and 19 in profiles/r3_envelope.log (1.22, 1.19); the suite's wall time is CPU-oracle time
def test_gradient_envelope_ratio_over_seeds(seed):
    """The gradient gates of test_parity_r2_gpu._grad_report compare the HIP path's distance to the fp32 oracle with the distance of the
    rounding-emulating twin to the same oracle (the envelope).  Both are realisations of one re-routing process, so their RATIO
    fluctuates from problem to problem and from build to build; this runs the gates on three more seeded problems (pinned shape P)
    and prints the ratio, so that the factors in those gates rest on more than the two problems of round 2 (which they did not survive:
    profiles/r3_envelope.log).  What bounds the gradients ABSOLUTELY is FP32_L2_CAP / FP32_COS_MIN there and the trajectory test above."""
    from oracle import avse_ref_cpu as orc
    model, twin, (x_a, x_v, y_a, y_v) = _build(2, 8, 256, 512, seed, precise=False, spatial_match="exact")
    shapes = (twin.stft_shape, twin.frame_shape, twin.output_stft_frames)
    emu = orc.AVFusionFramesRef(*shapes, spatial_match="exact", emulate_16bit=True)
    orc.load_seeded(emu, seed)
    emu.train()
    loss_ref, _, _, (a_ref, _, _) = orc.loss_ref(twin, x_a, x_v, y_a, y_v, 0.001, 1)
    loss_ref.backward()
    loss_emu, _, _, _ = orc.loss_ref(emu, x_a, x_v, y_a, y_v, 0.001, 1)
    loss_emu.backward()
    a, v, _ = model(x_a.cuda(), x_v.cuda())
    (F.mse_loss(a, y_a.cuda()) + 0.001 * F.mse_loss(v, y_v.cuda())).backward()
    mse = float(((a.detach().cpu() - a_ref.detach()) ** 2).mean())
    assert mse <= 1e-5, mse
    ratio = _grad_report(model, emu, twin, f"P shape, seed {seed}")
    assert ratio <= 2.5


@pytest.mark.parametrize("model_seed,vit_seed,frame_seed", [(101, 11, 21), pytest.param(57, 5, 33, marks=pytest.mark.slow),
                                                            pytest.param(71, 13, 17, marks=pytest.mark.slow)])      # the worst of four seed sets (5.7e-3); (57, 5, 33): 4.1e-3, (71, 13, 17): 3.0e-3 (profiles/r3_fp8_seeds.log)
def test_fp8_attention_end_to_end_bound_over_seeds(model_seed, vit_seed, frame_seed):
    """BASELINE config[4] (block-scaled fp8 Q K^T / P V inside the IEEE-half extractor): the end-to-end mask-MSE bound of
    tests/test_parity_r2_gpu.py (1e-2, one seed set: 3.9e-3) on three more seed sets of the P shape."""
    import maavss_amd
    from oracle import avse_ref_cpu as orc, vit_ref_cpu as vref
    from test_parity_r2_gpu import FP8_MASK_MSE_BOUND
    b, t, w = 2, 8, 256
    model, twin, (x_a, _, y_a, _) = _build(b, t, w, 512, model_seed, precise=False, spatial_match="exact")
    sd = vref.seeded_vit_state(vit_seed)
    va = maavss_amd.VideoAttention(path_to_weights="/nonexistent.pth", act_dtype="f16", attn_dtype="fp8")
    va.load_state_dict(sd)
    frames = vref.synthetic_frames(b * t, w, frame_seed)
    with torch.no_grad():
        x_v_ref = torch.stack([vref.clip_normalise_ref(vref.inference_ref(sd, frames[i * t:(i + 1) * t])) for i in range(b)])
    x_v = va.attention_frames(frames.cuda(), clip_frames=t).view(b, 1, t, w, w)
    _, _, _, (a_ref, _, _) = orc.loss_ref(twin, x_a, x_v_ref, y_a, x_v_ref[:, :, t // 2], 0.001, 1)
    a, _, _ = model(x_a.cuda(), x_v)
    mse = float(((a.detach().cpu() - a_ref.detach()) ** 2).mean())
    map_err = (x_v.cpu() - x_v_ref).abs()
    print(f"[parity] end to end, fp8 attention, seeds ({model_seed}, {vit_seed}, {frame_seed}): maps max|err| {map_err.max().item():.3e} mean "
          f"{map_err.mean().item():.3e}; mask-MSE {mse:.3e}")
    assert mse <= FP8_MASK_MSE_BOUND, mse
