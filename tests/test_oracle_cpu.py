"""CPU tests of the oracle itself: against the committed golden vectors (which were produced by the
reference's own AV_Fusion_Model_Frames, see oracle/make_golden.py) and against independent restatements."""
import os

import numpy as np
import pytest
import torch

from oracle import avse_ref_cpu as orc
from oracle import stft_ref_cpu as sref
from oracle import vit_ref_cpu as vref
from oracle import avfm_ref_cpu as avfm


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, f"avse_{name}.npz"), allow_pickle=False)
    meta = {k[5:]: z[k].item() for k in z.files if k.startswith("meta_")}
    return z, meta


@pytest.mark.parametrize("name", ["S", "P"])
def test_avse_oracle_matches_reference_golden(golden_dir, name):
    z, m = _load(golden_dir, name)
    n_bins = m["fft_len"] // 2 + 1
    t_a = m["hops_per_frame"] * m["frames"]
    model = orc.AVFusionFramesRef([m["batch"], 2, t_a, n_bins], [m["batch"], 1, m["frames"], m["width"], m["width"]],
                                  m["hops_per_frame"])
    orc.load_seeded(model, m["seed"])
    model.train()
    batch = orc.synthetic_batch(m["batch"], m["frames"], m["width"], t_a, n_bins, m["hops_per_frame"], m["seed"] + 1)
    loss, a_loss, v_loss, (a, v, fused) = orc.loss_ref(model, *batch, m["loss_coeff"], 1)
    loss.backward()
    np.testing.assert_allclose(a.detach().numpy(), z["x_a_out"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(fused.detach().numpy(), z["x_av_fused"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(v.detach().flatten()[::997].numpy(), z["x_v_out_sample"], rtol=1e-4, atol=2e-6)
    assert abs(loss.item() - z["loss"]) < 1e-6
    names = list(z["param_names"])
    params = dict(model.named_parameters())
    for i, k in enumerate(names):
        g = params[k].grad
        if z["grad_norm"][i] < 0:
            assert g is None, k
        else:
            assert abs(g.double().norm().item() - z["grad_norm"][i]) <= 1e-4 * z["grad_norm"][i] + 1e-9, k


@pytest.mark.parametrize("name", ["S", "P"])
def test_autoencoder_oracle_matches_reference_golden(golden_dir, name):
    """audio_ae_forward (stft_encoder -> ConvTranspose2d decoder) + mse + backward, as train_audio_net.py:107-109."""
    z = np.load(os.path.join(golden_dir, f"avse_ae_{name}.npz"), allow_pickle=False)
    m = {k[5:]: z[k].item() for k in z.files if k.startswith("meta_")}
    n_bins, t_a = m["fft_len"] // 2 + 1, m["hops_per_frame"] * m["frames"]
    model = orc.AVFusionFramesRef([m["batch"], 2, t_a, n_bins], [m["batch"], 1, m["frames"], m["width"], m["width"]],
                                  m["hops_per_frame"])
    orc.load_seeded(model, m["seed"])
    model.train()
    y = orc.synthetic_batch(m["batch"], m["frames"], m["width"], t_a, n_bins, m["hops_per_frame"], m["seed"] + 1)[0]
    yh = model.audio_ae_forward(y)
    assert yh.shape == y.shape
    loss = torch.nn.functional.mse_loss(yh, y)
    loss.backward()
    np.testing.assert_allclose(yh.detach().flatten()[::61].numpy(), z["yh_sample"], rtol=1e-4, atol=2e-6)
    assert abs(loss.item() - z["loss"]) < 1e-6
    params = dict(model.named_parameters())
    for i, k in enumerate(z["param_names"]):
        gn = params[str(k)].grad.double().norm().item()
        assert abs(gn - z["grad_norm"][i]) <= 1e-4 * z["grad_norm"][i] + 1e-9, k


def test_phasegram_variant_oracle_matches_reference_golden(golden_dir):
    """AVFusionRef + video_phasegram_ref reproduce what the reference's AV_Fusion_Model and
    utilities.video_phasegram produced in the build container (oracle/make_golden.py avfm)."""
    z = np.load(os.path.join(golden_dir, "avfm_A.npz"), allow_pickle=False)
    m = {k[5:]: z[k].item() for k in z.files if k.startswith("meta_")}
    b, t_a, n_bins, t, p = m["batch"], m["t_a"], m["n_bins"], m["frames"], m["p_size"]
    model = avfm.AVFusionRef([b, 2, t_a, n_bins], [b, 1, t, p * p], 8)
    avfm.load_seeded(model, m["seed"])
    g = torch.Generator().manual_seed(m["seed"] + 5)
    attn = torch.rand(b, 1, t, p, p, generator=g)
    x_v = avfm.video_phasegram_ref(attn)
    np.testing.assert_allclose(x_v.flatten()[::13].numpy(), z["pgram_sample"], rtol=0, atol=1e-6)
    assert abs(x_v.double().abs().sum().item() - z["pgram_abs_sum"]) < 1e-3
    x_a = torch.randn(b, 2, t_a, n_bins, generator=g) * 0.5
    y_a = torch.randn(b, 2, t_a, n_bins, generator=g) * 0.3
    model.train()
    yh_a, yh_v, fused = model(x_a, x_v)
    loss = torch.nn.functional.mse_loss(yh_v, x_v) + torch.nn.functional.mse_loss(yh_a, y_a)
    loss.backward()
    np.testing.assert_allclose(fused.detach().numpy(), z["full_fused"], rtol=1e-4, atol=2e-6)
    assert abs(loss.item() - z["full_loss"]) < 1e-6
    params = dict(model.named_parameters())
    for i, k in enumerate(z["full_param_names"]):
        gn = params[str(k)].grad.double().norm().item()
        assert abs(gn - z["full_grad_norm"][i]) <= 1e-4 * z["full_grad_norm"][i] + 1e-9, k


def test_constructor_shapes_and_guards():
    with pytest.raises(ValueError):
        orc.AVFusionFramesRef([1, 2, 128, 257], [1, 1, 16, 224, 224], 8)      # reference would loop forever
    with pytest.raises(ValueError):
        orc.AVFusionFramesRef([1, 2, 64, 257], [1, 1, 8, 256, 256], 8, latent_channels=64)
    m = orc.AVFusionFramesRef([1, 2, 128, 257], [1, 1, 16, 224, 224], 8, spatial_match="adaptive")
    a, v, f = m(torch.zeros(2, 2, 128, 257), torch.zeros(2, 1, 16, 224, 224))
    assert a.shape == (2, 2, 8, 257) and v.shape == (2, 1, 224, 224) and f.shape == (2, 512)
    assert orc.visual_side(256) == 4 and orc.visual_side(224) == 3 and orc.visual_side(384) == 6


def test_hop_size():
    assert sref.calc_hop_size(8, 8, 30, 16000) == (66, 4224, 64)
    assert sref.calc_hop_size(16, 8, 30, 16000) == (66, 8448, 128)


@pytest.mark.parametrize("fft_len", [256, 512, 1024])
def test_stft_oracle_vs_direct_dft(fft_len):
    audio = sref.synthetic_audio(3, 4224, 5)
    y = sref.stft_ref(audio, fft_len, 66)
    yd = sref.stft_direct_f64(audio, fft_len, 66)
    assert y.shape == (3, 2, 64, fft_len // 2 + 1)
    np.testing.assert_allclose(y.numpy(), yd.numpy(), rtol=0, atol=2e-6)
    w = torch.hamming_window(fft_len)
    np.testing.assert_allclose(w.numpy(), sref.hamming_periodic(fft_len).numpy(), atol=3e-7)


def test_istft_oracle_vs_direct_overlap_add():
    """istft_ref (torch.istft, what AV_Dataset.istft calls) against a float64 irfft + overlap-add written out by hand."""
    fft_len, hop, frames = 256, 66, 20
    g = torch.Generator().manual_seed(2)
    spec = torch.randn(2, 2, frames, fft_len // 2 + 1, generator=g)
    got = sref.istft_ref(spec, fft_len, hop, normalized=True)
    w = sref.hamming_periodic(fft_len, torch.float64)
    z = torch.complex(spec[:, 0].double(), spec[:, 1].double())                       # [B, T, F]
    fr = torch.fft.irfft(z * fft_len ** 0.5, n=fft_len, dim=-1) * w                  # normalized: * sqrt(n_fft)
    total = hop * (frames - 1) + fft_len
    num = torch.zeros(2, total, dtype=torch.float64)
    den = torch.zeros(total, dtype=torch.float64)
    for t in range(frames):
        num[:, t * hop:t * hop + fft_len] += fr[:, t]
        den[t * hop:t * hop + fft_len] += w * w
    want = (num / den)[:, fft_len // 2:fft_len // 2 + hop * (frames - 1)]
    np.testing.assert_allclose(got.double().numpy(), want.numpy(), rtol=0, atol=1e-5 * float(want.abs().max()))


def test_vit_oracle_vs_hf_vit():
    """Independent cross-check of the restated ViT-S/8 against transformers.ViTModel (local config only)."""
    tr = pytest.importorskip("transformers")
    cfg = tr.ViTConfig(hidden_size=384, num_hidden_layers=12, num_attention_heads=6, intermediate_size=1536,
                       image_size=224, patch_size=8, layer_norm_eps=1e-6, hidden_act="gelu", qkv_bias=True,
                       attn_implementation="eager")
    hf = tr.ViTModel(cfg, add_pooling_layer=False).eval()
    sd = vref.seeded_vit_state(3)
    m = {"embeddings.cls_token": sd["cls_token"], "embeddings.position_embeddings": sd["pos_embed"],
         "embeddings.patch_embeddings.projection.weight": sd["patch_embed.proj.weight"],
         "embeddings.patch_embeddings.projection.bias": sd["patch_embed.proj.bias"],
         "layernorm.weight": sd["norm.weight"], "layernorm.bias": sd["norm.bias"]}
    new_names = any(k.startswith("layers.0.attention.q_proj") for k in hf.state_dict())   # transformers >= 5
    for i in range(12):
        p = f"blocks.{i}."
        qw, qb = sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]
        if new_names:
            h = f"layers.{i}."
            qkv_names = [h + "attention.q_proj", h + "attention.k_proj", h + "attention.v_proj"]
            o, f1, f2 = h + "attention.o_proj", h + "mlp.fc1", h + "mlp.fc2"
        else:
            h = f"encoder.layer.{i}."
            qkv_names = [h + f"attention.attention.{nm}" for nm in ("query", "key", "value")]
            o, f1, f2 = h + "attention.output.dense", h + "intermediate.dense", h + "output.dense"
        for j, nm in enumerate(qkv_names):
            m[nm + ".weight"] = qw[384 * j:384 * (j + 1)]
            m[nm + ".bias"] = qb[384 * j:384 * (j + 1)]
        m[o + ".weight"], m[o + ".bias"] = sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"]
        m[h + "layernorm_before.weight"], m[h + "layernorm_before.bias"] = sd[p + "norm1.weight"], sd[p + "norm1.bias"]
        m[h + "layernorm_after.weight"], m[h + "layernorm_after.bias"] = sd[p + "norm2.weight"], sd[p + "norm2.bias"]
        m[f1 + ".weight"], m[f1 + ".bias"] = sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]
        m[f2 + ".weight"], m[f2 + ".bias"] = sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"]
    missing = hf.load_state_dict(m, strict=False)
    assert not missing.unexpected_keys and not missing.missing_keys, missing
    frames = vref.synthetic_frames(1, 224, 11)
    with torch.no_grad():
        ours = vref.get_last_selfattention(sd, frames)
        theirs = hf(pixel_values=frames, output_attentions=True).attentions[-1]
    np.testing.assert_allclose(ours.numpy(), theirs.numpy(), rtol=2e-3, atol=2e-6)


def test_attention_frames_postprocess():
    sd = vref.seeded_vit_state(3)
    frames = vref.synthetic_frames(2, 64, 12)
    with torch.no_grad():
        out = vref.inference_ref(sd, frames)
    assert out.shape == (2, 1, 64, 64)
    assert torch.allclose(out.flatten(1).max(1).values, torch.ones(2))
    clip = vref.clip_normalise_ref(out)
    assert clip.shape == (1, 2, 64, 64) and abs(clip.max().item() - 1) < 1e-6
    # nearest x8 upsample: 8x8 blocks are constant
    assert torch.equal(out[0, 0, :8, :8], out[0, 0, 0, 0].expand(8, 8))
