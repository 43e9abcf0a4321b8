#!/usr/bin/env python
"""The reference's training loop (train_avse_frames.py:112-205) on the drop-in classes, with synthetic clips instead of the MUSICES
data loader (no datasets in this environment): frames -> VideoAttention (ViT-S/8, HIP) -> attention frames; audio -> STFT (+ noise);
sliding windows of `num_frames` frames through AV_Fusion_Model_Frames; loss / backward / Adam as the reference does them; a checkpoint
written and re-loaded through the reference's own file layout (utilities.py:162-204).  Extraction of the next batch runs on a second HIP
stream (ClipPipeline).

    python examples/train_synthetic.py [--steps 6] [--batch 4] [--num_frames 8] [--num_seq 3] [--framesize 256]
    python examples/train_synthetic.py --overfit --steps 300 --lr 1e-4 --log_every 20     # one fixed batch: the loss has to fall
"""
import argparse
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maavss_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--num_frames", type=int, default=8)        # run_config.py: frames per window
    ap.add_argument("--num_seq", type=int, default=3)           # windows per optimizer step (train_avse_frames.py:143)
    ap.add_argument("--framesize", type=int, default=256)
    ap.add_argument("--fft_len", type=int, default=512)
    ap.add_argument("--hops_per_frame", type=int, default=8)
    ap.add_argument("--lr", type=float, default=1e-5)
    ap.add_argument("--overfit", action="store_true", help="train on ONE fixed batch (same clips, same noise draw) every step")
    ap.add_argument("--log_every", type=int, default=1)
    ap.add_argument("--precise", action="store_true", help="exact-f32 conv path instead of the 16-bit MFMA modes")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    b, nf, ns, w, hpf = a.batch, a.num_frames, a.num_seq, a.framesize, a.hops_per_frame
    t_total = nf + ns - 1                                      # frames per clip so that num_seq windows fit (av_dataset.py:251-278)
    hop, length, t_a = maavss_amd.calc_hop_size(t_total, hpf, 30, 16000)
    n_bins = a.fft_len // 2 + 1

    extractor = maavss_amd.VideoAttention(path_to_weights="dino_deitsmall8_pretrain.pth")      # random init when absent (no network)
    stft = maavss_amd.STFT(a.fft_len, hop, noise_std=0.1, device=dev)
    model = maavss_amd.AV_Fusion_Model_Frames([b, 2, hpf * nf, n_bins], [b, 1, nf, w, w], hpf, precise=a.precise).to(dev).train()
    step = maavss_amd.TrainStep(model, lr=a.lr, loss_coeff=0.001, num_seq=ns)
    pipe = maavss_amd.ClipPipeline(extractor, stft, clip_frames=t_total)

    g = torch.Generator().manual_seed(0)

    def batch():
        frames = torch.rand(b * t_total, 3, w, w, generator=g)
        audio = (0.3 * torch.randn(b, length, generator=g)).clamp(-1, 1)
        return frames.to(dev), audio.to(dev)

    fixed = batch() if a.overfit else None
    pipe.submit(*(fixed or batch()), seed=0)
    for i in range(a.steps):
        pipe.submit(*(fixed or batch()), seed=0 if a.overfit else i + 1)      # extraction of the next batch: side stream
        attn, x_stft, y_stft = pipe.get()                        # [B,1,T,H,W], [B,2,T_a,F] x 2
        losses = step.sliding_window_step(x_stft, y_stft, attn, attn, nf, hpf)
        pipe.release()
        if i % a.log_every == 0 or i == a.steps - 1:
            print(f"step {i}: a_loss {losses[0].item():.5f}  v_loss {losses[1].item():.5f}  loss {losses[2].item():.5f}", flush=True)
    pipe.drain()

    with tempfile.TemporaryDirectory() as cp_dir:
        maavss_amd.save_checkpoint(model.state_dict(), step.opt.state_dict(), 0, losses[2].item(), "synthetic", cp_dir)
        fresh = maavss_amd.AV_Fusion_Model_Frames([b, 2, hpf * nf, n_bins], [b, 1, nf, w, w], hpf).to(dev)
        opt = maavss_amd.FusedAdam(fresh, lr=a.lr)
        maavss_amd.load_checkpoint(fresh, opt, cp_dir, auto=True, load_opt=True)
        same = all(torch.equal(p, q) for p, q in zip(model.state_dict().values(), fresh.state_dict().values()))
        print("checkpoint round trip:", "ok" if same else "MISMATCH")
        if not same:
            sys.exit(1)


if __name__ == "__main__":
    main()
