"""Achieved HBM GB/s of the STFT + noise kernel (K17, `maavss_stft_fwd`) over the clip batch size (VERDICT r2 item 7 / weak 12:
at B = 32 the launch is one wave round of the chip and latency-bound; what does the kernel reach when it has work?).
Algorithmic bytes per clip: 4 L in + 2 x (2 * T_a * F * 4) out.  GPU box only."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maavss_amd  # noqa: E402

fft, t, hpf = 512, 16, 8
hop, length, t_a = maavss_amd.calc_hop_size(t, hpf, 30, 16000)
f = fft // 2 + 1
stft = maavss_amd.STFT(fft, hop, noise_std=0.1, device="cuda")
rows = []
for b in (32, 256, 2048, 8192):
    audio = torch.randn(b, length, device="cuda").clamp_(-1, 1)
    for _ in range(3):
        stft(audio, seed=1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    torch.cuda.synchronize()
    e0.record()
    for i in range(n):
        x, y = stft(audio, seed=i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    by = b * (4.0 * length + 2 * 2 * t_a * f * 4.0)
    rows.append(dict(batch=b, us_per_launch=round(us, 1), algorithmic_MB=round(by / 1e6, 2), GBps=round(by / us / 1e3, 1),
                     hbm_frac=round(by / us / 1e3 / 8000.0, 3)))
    del x, y, audio
print(json.dumps({"kernel": "maavss_stft_fwd (512-pt, T=16, hop 66)", "note": "back-to-back launches incl. the two torch.empty of the outputs", "rows": rows}, indent=1))
