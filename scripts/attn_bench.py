#!/usr/bin/env python
"""Micro-benchmark of maavss_vit_attn on the GPU box (HIP events on the launch stream):
    python scripts/attn_bench.py [--frames 512] [--ntok 785] [--dtype 0|2|3|4] [--iters 20]
Prints avg launch time, TFLOP/s on the useful 4*frames*heads*ntok^2*64 FLOPs and the fraction of the 2.5 PFLOP/s bf16 peak.
MAAVSS_ATTN_VARIANT=1 selects the round-1 kernel (bf16) for A/B runs in separate processes."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maavss_amd import _lib  # noqa: E402

if os.environ.get("MAAVSS_LIB"):          # measurement builds (make -C maavss_amd/csrc ablate)
    _lib.LIB_PATH = os.environ["MAAVSS_LIB"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--ntok", type=int, default=785)
    ap.add_argument("--dtype", type=int, default=0)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--zeros", action="store_true")
    a = ap.parse_args()
    rows = a.frames * a.ntok
    fp8 = a.dtype == 3            # fp8 attention on f16 qkv / out
    mx = a.dtype == 4             # block-scaled fp8 (MX) attention kernel alone: the images are quantised once, outside the timing
    tdt = {0: torch.bfloat16, 2: torch.float16, 3: torch.float16, 4: torch.float16}[a.dtype]
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv = torch.randn(rows, 1152, device="cuda", generator=g)
    qkv[:, :384] *= 0.125 * 1.4426950408889634
    if a.zeros:
        qkv.zero_()
    qkv = qkv.to(tdt)
    out = torch.empty(rows, 384, device="cuda", dtype=tdt)
    st = _lib.stream_ptr()
    if mx:
        ws = torch.empty(_lib.query("maavss_vit_attn_mx_ws_bytes", rows), device="cuda", dtype=torch.uint8)
        _lib.call("maavss_vit_qkv_mx", qkv.data_ptr(), ws.data_ptr(), rows, 1152, 2, st)
        run = lambda: _lib.call("maavss_vit_attn_mx", ws.data_ptr(), out.data_ptr(), a.frames, a.ntok, 6, 384, 2, st)
    elif fp8:
        raise SystemExit("--dtype 3 (round 2's non-scaled fp8 kernel) was removed in round 3: use --dtype 4 (block-scaled fp8)")
    else:
        run = lambda: _lib.call("maavss_vit_attn", qkv.data_ptr(), out.data_ptr(), a.frames, a.ntok, 6, 1152, 384, a.dtype, st)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3
    fl = 4.0 * a.frames * 6 * a.ntok * a.ntok * 64
    tf = fl / (us * 1e-6) / 1e12
    print(f"vit_attn variant={os.environ.get('MAAVSS_ATTN_VARIANT', 'default')} dtype={a.dtype} frames={a.frames} ntok={a.ntok}: "
          f"{us:.1f} us/launch, {tf:.0f} TFLOP/s = {tf / 2500:.3f} of bf16 peak")


if __name__ == "__main__":
    main()
