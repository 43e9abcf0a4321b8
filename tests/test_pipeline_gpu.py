"""ClipPipeline (maavss_amd/pipeline.py): attention-frame extraction + STFT of batch i+1 on a side HIP stream under the training
step of batch i -- the reference's data path (av_dataset.py:321,335-342) has no dependency on the optimizer step
(train_avse_frames.py:150-181).  The pipelined loop must produce what the serial loop produces: the extractor's outputs bit for bit
(same kernels, another stream) and -- in deterministic mode (maavss_amd.set_deterministic: no f32-atomic split-K) -- the training
losses and the weights bit for bit too, over more batches than the pipeline has slots: a missing event or a slot reused too early
shows up as a mismatch."""
import pytest
import torch

pytestmark = pytest.mark.gpu

B, T, W, FFT, HPF, NBATCH = 2, 8, 128, 256, 8, 5


def _setup(seed):
    import maavss_amd
    from oracle import avse_ref_cpu as orc, stft_ref_cpu as sref, vit_ref_cpu as vref
    hop, length, t_a = maavss_amd.calc_hop_size(T, HPF, 30, 16000)
    n_bins = FFT // 2 + 1
    shapes = ([B, 2, t_a, n_bins], [B, 1, T, W, W], HPF)
    model = maavss_amd.AV_Fusion_Model_Frames(*shapes)
    model.load_state_dict(orc.seeded_state_dict(orc.AVFusionFramesRef(*shapes), seed), strict=True)
    model = model.cuda().train()
    va = maavss_amd.VideoAttention(path_to_weights="/nonexistent.pth")
    va.load_state_dict(vref.seeded_vit_state(3))
    stft = maavss_amd.STFT(FFT, hop, noise_std=0.1, device="cuda")
    step = maavss_amd.TrainStep(model, lr=1e-4)
    frames = [vref.synthetic_frames(B * T, W, 100 + i).cuda() for i in range(NBATCH)]
    audio = [sref.synthetic_audio(B, length, 200 + i).cuda() for i in range(NBATCH)]
    return maavss_amd, model, va, stft, step, frames, audio


def _train(step, x_v, x_stft, y_stft):
    mid = T // 2
    return step(x_stft, x_v, y_stft[:, :, mid * HPF:(mid + 1) * HPF, :], x_v[:, :, mid])[2].item()


def test_pipelined_loop_equals_the_serial_loop():
    import maavss_amd as _m
    prev = _m.set_deterministic(True)
    try:
        _pipelined_vs_serial()
    finally:
        _m.set_deterministic(prev)


def _pipelined_vs_serial():
    maavss_amd, model, va, stft, step, frames, audio = _setup(61)
    serial_losses, serial_attn, serial_stft = [], [], []
    for i in range(NBATCH):
        attn = va.attention_frames(frames[i], clip_frames=T).view(B, 1, T, W, W)
        x_stft, y_stft = stft(audio[i], seed=i)
        serial_attn.append(attn.clone())
        serial_stft.append((x_stft.clone(), y_stft.clone()))
        serial_losses.append(_train(step, attn, x_stft, y_stft))
    torch.cuda.synchronize()
    w_serial = step.flat.params.clone()

    maavss_amd, model, va, stft, step, frames, audio = _setup(61)
    pipe = maavss_amd.ClipPipeline(va, stft, T)
    pipe.submit(frames[0], audio[0], seed=0)
    piped_losses = []
    for i in range(NBATCH):
        if i + 1 < NBATCH:
            pipe.submit(frames[i + 1], audio[i + 1], seed=i + 1)
        x_v, x_stft, y_stft = pipe.get()
        assert torch.equal(x_v, serial_attn[i]), f"batch {i}: attention frames differ from the serial loop"
        assert torch.equal(x_stft, serial_stft[i][0]) and torch.equal(y_stft, serial_stft[i][1]), f"batch {i}: STFT differs"
        piped_losses.append(_train(step, x_v, x_stft, y_stft))
        pipe.release()
    pipe.drain()
    assert serial_losses == piped_losses, (serial_losses, piped_losses)
    assert torch.equal(step.flat.params, w_serial)
    print(f"[pipeline] {NBATCH} batches: losses {piped_losses[0]:.6f} .. {piped_losses[-1]:.6f}; losses and weights bit-identical to the serial loop")
    with pytest.raises(AssertionError):
        pipe.get()                                   # nothing submitted
    # inputs that the caller frees right after submit() (temporaries) must stay valid for the side stream: same values, fresh
    # tensors, references dropped at once, allocator pressure in between
    maavss_amd, model, va, stft, step, frames, audio = _setup(61)
    pipe = maavss_amd.ClipPipeline(va, stft, T)
    for i in range(3):
        pipe.submit(frames[i].clone(), audio[i].clone(), seed=i)
        junk = [torch.full_like(frames[i], float("nan")) for _ in range(2)]      # re-uses the freed blocks if they were released
        del junk
        if i >= 1:
            x_v, x_stft, y_stft = pipe.get()
            assert torch.equal(x_v, serial_attn[i - 1]) and torch.equal(y_stft, serial_stft[i - 1][1]), f"temporary inputs, batch {i - 1}"
            pipe.release()
    pipe.drain()
