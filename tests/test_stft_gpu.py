"""GPU parity: K17 STFT(+noise) kernel through the C-ABI vs the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fft_len,frames", [(512, 8), (256, 8), (1024, 32), (512, 16)])
def test_stft_matches_oracle(fft_len, frames):
    import maavss_amd
    from oracle import stft_ref_cpu as sref
    hop, length, t_a = maavss_amd.calc_hop_size(frames, 8, 30, 16000)
    audio = sref.synthetic_audio(3, length, 5)
    g = torch.Generator().manual_seed(9)
    noise = torch.randn(3, 2, t_a, fft_len // 2 + 1, generator=g)
    st = maavss_amd.STFT(fft_len, hop, noise_std=0.1, device="cuda")
    x, y = st(audio.cuda(), noise=noise.cuda())
    xr, yr = sref.gen_stft_example_ref(audio, fft_len, hop, 0.1, noise)
    assert y.shape == (3, 2, t_a, fft_len // 2 + 1)
    np.testing.assert_allclose(y.cpu().numpy(), yr.numpy(), rtol=0, atol=5e-6)   # fp32 tolerance, |y| <= ~0.5
    np.testing.assert_allclose(x.cpu().numpy(), xr.numpy(), rtol=0, atol=5e-6)


def test_stft_trim_and_normalise_output():
    import maavss_amd
    from oracle import stft_ref_cpu as sref
    hop, length, t_a = maavss_amd.calc_hop_size(8, 8, 30, 16000)
    audio = sref.synthetic_audio(2, length, 6)
    noise = torch.randn(2, 2, t_a, 257, generator=torch.Generator().manual_seed(1))
    st = maavss_amd.STFT(512, hop, noise_std=0.1, normalize_output_fft=True, device="cuda")
    x, y = st(audio.cuda(), noise=noise.cuda())
    xr, yr = sref.gen_stft_example_ref(audio, 512, hop, 0.1, noise, normalize_output=True)
    np.testing.assert_allclose(y.cpu().numpy(), yr.numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(x.cpu().numpy(), xr.numpy(), rtol=0, atol=2e-5)
    st2 = maavss_amd.STFT(512, hop, trim_stft_end=True, device="cuda")
    _, y2 = st2(audio.cuda(), want_x=False)
    np.testing.assert_allclose(y2.cpu().numpy(), sref.stft_ref(audio, 512, hop, trim_stft_end=True).numpy(), atol=5e-6)


def test_stft_device_noise_statistics():
    """In-kernel Philox noise: x - y must be N(0, sigma^2) (no reference RNG stream to match)."""
    import maavss_amd
    from oracle import stft_ref_cpu as sref
    hop, length, t_a = maavss_amd.calc_hop_size(16, 8, 30, 16000)
    audio = sref.synthetic_audio(8, length, 7).cuda()
    st = maavss_amd.STFT(512, hop, noise_std=0.1, device="cuda")
    x, y = st(audio, seed=123)
    d = (x - y).flatten().double() / 0.1
    assert abs(d.mean().item()) < 5e-3 and abs(d.std().item() - 1) < 5e-3
    assert abs((d ** 3).mean().item()) < 2e-2 and abs((d ** 4).mean().item() - 3) < 5e-2
    x2, _ = st(audio, seed=124)
    assert (x2 - x).abs().max().item() > 0.1      # different seed, different draw
    x3, _ = st(audio, seed=123)
    assert torch.equal(x3, x)                     # same seed, same draw


def test_stft_device_noise_last_bin_and_grid_independence():
    """The bin n_fft / 2 takes its noise from a per-PAIR Philox block that a wave evaluates for its next 64 pairs at once (stft.hip): the
    draw must not depend on how the pairs are dealt to the waves (a 256-clip launch walks two pairs per wave, an 8-clip launch one), the
    last bin must be N(0, sigma^2) like the others, and no two frames may share its values."""
    import maavss_amd
    from oracle import stft_ref_cpu as sref
    hop, length, t_a = maavss_amd.calc_hop_size(16, 8, 30, 16000)
    audio = sref.synthetic_audio(264, length, 7).cuda()
    for fft_len in (512, 256, 1024):
        st = maavss_amd.STFT(fft_len, hop, noise_std=0.1, device="cuda")
        x_big, y_big = st(audio, seed=5)
        x_small, y_small = st(audio[:8].contiguous(), seed=5)
        assert torch.equal(y_big[:8], y_small)
        assert torch.equal(x_big[:8], x_small), fft_len
        d = ((x_big - y_big)[..., fft_len // 2] / 0.1).double()          # [264, 2, t_a]: the last bin, re and im planes
        assert abs(d.mean().item()) < 2e-2 and abs(d.std().item() - 1) < 2e-2, (fft_len, d.mean().item(), d.std().item())
        assert abs((d ** 4).mean().item() - 3) < 0.15
        flat = d.flatten()
        assert flat.unique().numel() > 0.999 * flat.numel()               # frames do not share draws
        # neighbouring frames (the two halves of a pair, and consecutive pairs) are uncorrelated
        c1 = (d[:, :, 1:] * d[:, :, :-1]).mean().item()
        assert abs(c1) < 2e-2, c1


@pytest.mark.parametrize("fft_len,hop,frames,trim,normalized", [(512, 66, 64, False, True), (256, 66, 64, False, True),
                                                                (512, 66, 128, True, True), (1024, 100, 40, False, False)])
def test_istft_matches_oracle(fft_len, hop, frames, trim, normalized):
    """STFT.inverse == AV_Dataset.istft restated on torch.istft (av_dataset.py:181-201): random spectra (not the STFT
    of anything, so the overlap-add / envelope arithmetic is exercised, not just a round trip) and a true round trip."""
    import maavss_amd
    from oracle import stft_ref_cpu as sref
    g = torch.Generator().manual_seed(11)
    f = fft_len // 2 + (0 if trim else 1)
    spec = torch.randn(3, 2, frames, f, generator=g)
    st = maavss_amd.STFT(fft_len, hop, normalized=normalized, trim_stft_end=trim)
    got = st.inverse(spec.cuda())
    want = sref.istft_ref(spec, fft_len, hop, normalized, trim)
    assert got.shape == want.shape == (3, hop * (frames - 1))
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=0, atol=2e-5 * float(want.abs().max()))
    one = st.inverse(spec[1].cuda())
    np.testing.assert_allclose(one.cpu().numpy(), want[1].numpy(), rtol=0, atol=2e-5 * float(want.abs().max()))
    # round trip through the HIP forward: the reference's own scaling mismatch (window energy vs sqrt(n_fft)) is a constant
    audio = sref.synthetic_audio(2, hop * frames, 4)
    _, y = st(audio.cuda(), want_x=False)
    back = st.inverse(y)
    ref_back = sref.istft_ref(sref.stft_ref(audio, fft_len, hop, normalized, trim), fft_len, hop, normalized, trim)
    np.testing.assert_allclose(back.cpu().numpy(), ref_back.numpy(), rtol=0, atol=3e-5 * float(ref_back.abs().max()))
