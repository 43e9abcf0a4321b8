"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/avse_*.npz.

Runs ONLY in the build container (needs /root/reference).  It imports the
reference's own `AV_Fusion_Model_Frames` (avse_model_final.py:14) through a
harness shim -- `torchsummary` (a print-only utility, avse_model_final.py:8)
stubbed, and the hard-coded device string "cuda" (avse_model_final.py:59,66,...)
mapped to "cpu" -- loads the seeded weight recipe of oracle/avse_ref_cpu.py into
it, runs forward / loss / backward / one Adam step on seeded synthetic inputs and
stores the results as small fixtures.  It also asserts that the oracle twin
(AVFusionFramesRef) reproduces the reference on the same inputs, which is what
pins the oracle.  Nothing of the reference (source or bytecode) is written out:
only numbers.

    python oracle/make_golden.py            # writes tests/golden/avse_{P,S,L}.npz and avse_ae_{P,S}.npz
    python oracle/make_golden.py ae         # only the autoencoder fixtures (audio_ae_forward)
    python oracle/make_golden.py avfm       # only the phasegram-variant fixtures (avse_model.AV_Fusion_Model, 8 f1)
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import avse_ref_cpu as orc  # noqa: E402

REFERENCE = "/root/reference"

# name -> (batch, T frames, W, fft_len, hops_per_frame)
CONFIGS = {
    "P": (2, 8, 256, 512, 8),      # parity config
    "S": (2, 8, 128, 256, 8),      # small: S=4, 3-layer STFT encoder
    "L": (2, 16, 256, 512, 8),     # BASELINE config-2 clip shape (T=16) at the parity frame size
}
SEED = 7
LR = 1e-3
LOSS_COEFF = 0.001   # run_config.py:8


@contextlib.contextmanager
def reference_on_cpu():
    """Make the reference importable/constructible without a GPU."""
    stub = types.ModuleType("torchsummary")
    stub.summary = lambda *a, **k: None
    sys.modules["torchsummary"] = stub
    sys.path.insert(0, REFERENCE)
    t_to, m_to = torch.Tensor.to, torch.nn.Module.to

    def fix(args):
        return tuple("cpu" if (isinstance(a, str) and a == "cuda") else a for a in args)

    torch.Tensor.to = lambda self, *a, **k: t_to(self, *fix(a), **k)
    torch.nn.Module.to = lambda self, *a, **k: m_to(self, *fix(a), **k)
    try:
        yield
    finally:
        torch.Tensor.to, torch.nn.Module.to = t_to, m_to
        sys.path.remove(REFERENCE)


def run(model, x_a, x_v, y_a, y_v):
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=LR)
    opt.zero_grad()
    loss, a_loss, v_loss, (a, v, fused) = orc.loss_ref(model, x_a, x_v, y_a, y_v, LOSS_COEFF, 1)
    loss.backward()
    out = {
        "x_a_out": a.detach().numpy(), "x_av_fused": fused.detach().numpy(),
        "x_v_out_sample": v.detach().flatten()[::997].numpy(),
        "x_v_out_sum": np.float64(v.detach().double().sum().item()),
        "loss": np.float64(loss.item()), "a_loss": np.float64(a_loss.item()), "v_loss": np.float64(v_loss.item()),
    }
    names, gnorm, gsample = [], [], []
    for k, p in model.named_parameters():
        if k.startswith("stft_autoencoder."):
            continue
        names.append(k)
        if p.grad is None:
            gnorm.append(-1.0)
            gsample.append(np.zeros(8, np.float32))
        else:
            g = p.grad.detach().flatten()
            gnorm.append(g.double().norm().item())
            idx = (torch.arange(8, dtype=torch.long) * (g.numel() - 1)) // 7
            gsample.append(g[idx].numpy())
    out["param_names"] = np.array(names)
    out["grad_norm"] = np.array(gnorm)
    out["grad_sample"] = np.stack(gsample)
    opt.step()
    wsum, wabs = [], []
    for k, p in model.named_parameters():
        if k.startswith("stft_autoencoder."):
            continue
        wsum.append(p.detach().double().sum().item())
        wabs.append(p.detach().double().abs().sum().item())
    out["adam_wsum"] = np.array(wsum)
    out["adam_wabs"] = np.array(wabs)
    bn_names, bn_mean, bn_var = [], [], []
    for k, b in model.named_buffers():
        if k.startswith("stft_autoencoder.") or k.startswith("stft_decoder."):
            continue
        if k.endswith("running_mean"):
            bn_names.append(k[:-len(".running_mean")])
            bn_mean.append(b.double().sum().item())
        if k.endswith("running_var"):
            bn_var.append(b.double().sum().item())
    out["bn_names"] = np.array(bn_names)
    out["bn_running_mean_sum"] = np.array(bn_mean)
    out["bn_running_var_sum"] = np.array(bn_var)
    return out


def run_ae(model, y_stft):
    """One step of the reference's autoencoder trainer (train_audio_net.py:97-110): yh = audio_ae_forward(y),
    loss = mse(yh, y), backward.  Full output, loss, every gradient's norm + 8 samples, decoder/encoder BN statistics."""
    model.train()
    for p in model.parameters():
        p.grad = None
    yh = model.audio_ae_forward(y_stft)
    loss = torch.nn.functional.mse_loss(yh, y_stft)
    loss.backward()
    out = {"yh_sample": yh.detach().flatten()[::61].numpy(), "yh_sum": np.float64(yh.detach().double().sum().item()),
           "yh_abs_sum": np.float64(yh.detach().double().abs().sum().item()), "loss": np.float64(loss.item())}
    names, gnorm, gsample = [], [], []
    for k, p in model.named_parameters():
        if not (k.startswith("stft_encoder.") or k.startswith("stft_decoder.")):
            continue
        names.append(k)
        g = p.grad.detach().flatten()
        gnorm.append(g.double().norm().item())
        idx = (torch.arange(8, dtype=torch.long) * (g.numel() - 1)) // 7
        gsample.append(g[idx].numpy())
    out["param_names"] = np.array(names)
    out["grad_norm"] = np.array(gnorm)
    out["grad_sample"] = np.stack(gsample)
    bn_names, bn_mean, bn_var = [], [], []
    for k, b in model.named_buffers():
        if not (k.startswith("stft_encoder.") or k.startswith("stft_decoder.")):
            continue
        if k.endswith("running_mean"):
            bn_names.append(k[:-len(".running_mean")])
            bn_mean.append(b.double().sum().item())
        if k.endswith("running_var"):
            bn_var.append(b.double().sum().item())
    out["bn_names"] = np.array(bn_names)
    out["bn_running_mean_sum"] = np.array(bn_mean)
    out["bn_running_var_sum"] = np.array(bn_var)
    return out


def _grad_summary(model, prefixes=None):
    names, gnorm, gsample = [], [], []
    for k, p in model.named_parameters():
        if prefixes is not None and not k.startswith(prefixes):
            continue
        names.append(k)
        if p.grad is None:
            gnorm.append(-1.0)
            gsample.append(np.zeros(8, np.float32))
            continue
        g = p.grad.detach().flatten()
        gnorm.append(g.double().norm().item())
        idx = (torch.arange(8, dtype=torch.long) * (g.numel() - 1)) // 7
        gsample.append(g[idx].numpy())
    return {"param_names": np.array(names), "grad_norm": np.array(gnorm), "grad_sample": np.stack(gsample)}


def _bn_summary(model):
    names, mean, var = [], [], []
    for k, b in model.named_buffers():
        if "autoencoder" in k:
            continue
        if k.endswith("running_mean"):
            names.append(k[:-len(".running_mean")])
            mean.append(b.double().sum().item())
        if k.endswith("running_var"):
            var.append(b.double().sum().item())
    return {"bn_names": np.array(names), "bn_running_mean_sum": np.array(mean), "bn_running_var_sum": np.array(var)}


def run_avfm(model, x_a, x_v, y_a):
    """train_av_net.py:121-131: yh_stft, yh_pgram, fused = model(x_stft, y_phasegram); a_loss = mse(yh_pgram, y_phasegram),
    v_loss = mse(yh_stft, y_stft) (the reference's names are swapped), loss = a_loss + v_loss; backward with every
    parameter trainable (a superset of the trainer's frozen-encoder setting)."""
    model.train()
    for p in model.parameters():
        p.grad = None
        p.requires_grad_(True)
    yh_a, yh_v, fused = model(x_a, x_v)
    loss = torch.nn.functional.mse_loss(yh_v, x_v) + torch.nn.functional.mse_loss(yh_a, y_a)
    loss.backward()
    out = {"a_sample": yh_a.detach().flatten()[::127].numpy(), "v_sample": yh_v.detach().flatten()[::31].numpy(),
           "fused": fused.detach().numpy(), "loss": np.float64(loss.item())}
    out.update(_grad_summary(model, ("phasegram_encoder.", "stft_encoder.", "lstm.", "fc1.", "fc2.", "a_fc1.", "v_fc1.")))
    out.update(_bn_summary(model))
    res = {"full_" + k: v for k, v in out.items()}
    # the two autoencoder entry points (train_visual_net / train_audio_net style steps)
    for tag, fn, x in (("vae", model.visual_ae_forward, x_v), ("aae", model.audio_ae_forward, x_a)):
        for p in model.parameters():
            p.grad = None
        yh = fn(x)
        loss = torch.nn.functional.mse_loss(yh, x)
        loss.backward()
        o = {"out_sample": yh.detach().flatten()[::53].numpy(), "loss": np.float64(loss.item())}
        o.update(_grad_summary(model, ("phasegram_encoder.", "phasegram_decoder.") if tag == "vae" else ("stft_encoder.", "stft_decoder.")))
        res.update({f"{tag}_" + k: v for k, v in o.items()})
    return res


def main_avfm():
    """pin oracle/avfm_ref_cpu.py (AVFusionRef, video_phasegram_ref) to the reference."""
    from oracle import avfm_ref_cpu as avfm
    with reference_on_cpu():
        with contextlib.redirect_stdout(io.StringIO()):
            import avse_model as ref_mod
        # utilities.py needs torchvision / cv2 / wandb (absent): video_phasegram is taken from its source text by name --
        # executed here, in the build container only, never copied into the repo
        import ast
        src = open(os.path.join(REFERENCE, "utilities.py")).read()
        fn_src = next(ast.get_source_segment(src, n) for n in ast.parse(src).body
                      if isinstance(n, ast.FunctionDef) and n.name == "video_phasegram")
        ns = {"torch": torch, "np": np}
        exec(compile(fn_src, "utilities.video_phasegram", "exec"), ns)
        b, t_a, n_bins, t, p = 2, 64, 256, 8, 32
        stft_shape, pgram_shape = [b, 2, t_a, n_bins], [b, 1, t, p * p]
        with contextlib.redirect_stdout(io.StringIO()):
            torch.manual_seed(0)
            ref = ref_mod.AV_Fusion_Model(stft_shape, pgram_shape, 8)
        twin = avfm.AVFusionRef(stft_shape, pgram_shape, 8)
        assert list(ref.state_dict().keys()) == list(twin.state_dict().keys()), "state_dict keys differ"
        sd = avfm.seeded_state_dict(twin, SEED)
        g = torch.Generator().manual_seed(SEED + 5)
        attn = torch.rand(b, 1, t, p, p, generator=g)
        x_v_ref = ns["video_phasegram"](attn.clone(), resize=None, diff=True, cumulative=True, normalize=True)
        x_v = avfm.video_phasegram_ref(attn)
        np.testing.assert_allclose(x_v.numpy(), x_v_ref.numpy(), rtol=0, atol=1e-6, err_msg="video_phasegram")
        x_a = torch.randn(b, 2, t_a, n_bins, generator=g) * 0.5
        y_a = torch.randn(b, 2, t_a, n_bins, generator=g) * 0.3
        ref.load_state_dict(sd, strict=True)
        twin.load_state_dict(sd, strict=True)
        got_ref, got_twin = run_avfm(ref, x_a, x_v, y_a), run_avfm(twin, x_a, x_v, y_a)
        for k in got_ref:
            if got_ref[k].dtype.kind in "US":
                assert (got_ref[k] == got_twin[k]).all(), k
            else:
                np.testing.assert_allclose(got_twin[k], got_ref[k], rtol=2e-5, atol=1e-6, err_msg=f"avfm:{k}")
        meta = dict(batch=b, t_a=t_a, n_bins=n_bins, frames=t, p_size=p, seed=SEED)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "avfm_A.npz"), attn_sample=attn.flatten()[::97].numpy(),
                            pgram_sample=x_v_ref.flatten()[::13].numpy(), pgram_abs_sum=np.float64(x_v_ref.double().abs().sum().item()),
                            **got_ref, **{f"meta_{k}": np.array(v) for k, v in meta.items()})
        print(f"[golden] avfm A: loss={got_ref['full_loss']:.8f} oracle==reference OK (model and video_phasegram)", flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "avfm":
        return main_avfm()
    only_ae = len(sys.argv) > 1 and sys.argv[1] == "ae"
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    torch.set_num_threads(8)
    with reference_on_cpu():
        with contextlib.redirect_stdout(io.StringIO()):
            import avse_model_final as ref_mod
        for name, (b, t, w, fft, hpf) in CONFIGS.items():
            n_bins = fft // 2 + 1
            t_a = hpf * t
            stft_shape = [b, 2, t_a, n_bins]
            frame_shape = [b, 1, t, w, w]
            with contextlib.redirect_stdout(io.StringIO()):
                torch.manual_seed(0)
                ref = ref_mod.AV_Fusion_Model_Frames(stft_shape, frame_shape, hpf)
            twin = orc.AVFusionFramesRef(stft_shape, frame_shape, hpf)
            assert list(ref.state_dict().keys()) == list(twin.state_dict().keys()), "state_dict keys differ"
            for k, v in ref.state_dict().items():
                assert v.shape == twin.state_dict()[k].shape, k
            ref.load_state_dict(orc.seeded_state_dict(twin, SEED), strict=True)
            orc.load_seeded(twin, SEED)
            batch = orc.synthetic_batch(b, t, w, t_a, n_bins, hpf, SEED + 1)
            meta = dict(batch=b, frames=t, width=w, fft_len=fft, hops_per_frame=hpf, seed=SEED,
                        lr=LR, loss_coeff=LOSS_COEFF)
            if name in ("P", "S"):
                # autoencoder step first (fresh BN buffers); the weights are re-seeded before the fusion step below
                y_stft = batch[0]          # the synthetic STFT [B,2,T_a,F] (the trainer feeds the clean STFT to the autoencoder)
                ae_ref, ae_twin = run_ae(ref, y_stft), run_ae(twin, y_stft)
                for k in ae_ref:
                    if ae_ref[k].dtype.kind in "US":
                        assert (ae_ref[k] == ae_twin[k]).all(), k
                    else:
                        np.testing.assert_allclose(ae_twin[k], ae_ref[k], rtol=2e-5, atol=1e-6, err_msg=f"ae {name}:{k}")
                np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"avse_ae_{name}.npz"),
                                    **ae_ref, **{f"meta_{k}": np.array(v) for k, v in meta.items()})
                print(f"[golden] ae {name}: loss={ae_ref['loss']:.8f} oracle==reference OK", flush=True)
                ref.load_state_dict(orc.seeded_state_dict(twin, SEED), strict=True)
                orc.load_seeded(twin, SEED)
            if only_ae:
                continue
            got_ref = run(ref, *batch)
            got_twin = run(twin, *batch)
            for k in got_ref:
                if got_ref[k].dtype.kind in "US":
                    assert (got_ref[k] == got_twin[k]).all(), k
                else:
                    np.testing.assert_allclose(got_twin[k], got_ref[k], rtol=2e-5, atol=1e-6, err_msg=f"{name}:{k}")
            np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"avse_{name}.npz"),
                                **got_ref, **{f"meta_{k}": np.array(v) for k, v in meta.items()})
            print(f"[golden] {name}: loss={got_ref['loss']:.8f} params={len(got_ref['param_names'])} "
                  f"oracle==reference OK", flush=True)
    if not only_ae:
        main_avfm()


if __name__ == "__main__":
    main()
