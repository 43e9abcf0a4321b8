"""Timing of the Conv3d weight-gradient launches of one training step at the benched shape (B=32, T=16, 224^2; 16-bit path: x f32, dy bf16)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maavss_amd import _lib
if os.environ.get("MAAVSS_LIB"):          # A/B against another build of the library
    _lib.LIB_PATH = os.environ["MAAVSS_LIB"]
from maavss_amd import ops


def timed(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


g = torch.Generator(device="cuda").manual_seed(1)
tot = 0.0
for name, ci, co, hw, pad in (("16->32 @112^2", 16, 32, 112, 2), ("32->64 @56^2", 32, 64, 56, 2), ("64->64 @28^2", 64, 64, 28, 2), ("64->16 @9^2", 64, 16, 9, 3)):
    x = torch.randn(32, 16, hw, hw, ci, device="cuda", generator=g)
    ho = hw + 2 * pad - 4
    dy = torch.randn(32, 16, ho, ho, co, device="cuda", generator=g).to(torch.bfloat16)
    dw = torch.empty(co, ci, 3, 5, 5, device="cuda")
    us = timed(lambda: ops.conv3d_wgrad(x, dy, pad, ops.MODE_BF16, dw=dw))
    ref = ops.conv3d_wgrad(x, dy.float(), pad, ops.MODE_F32)
    rel = ((dw - ref).norm() / ref.norm()).item()
    fl = 2.0 * 32 * 16 * ho * ho * 75 * ci * co
    extra = ""
    if (ci, co) in getattr(ops, "WGRAD_X16_SHAPES", ()):      # x as the bf16 copy its producer writes: LDS-DMA staging
        xb = x.bfloat16()
        dw16 = torch.empty_like(dw)
        us16 = timed(lambda: ops.conv3d_wgrad(xb, dy, pad, ops.MODE_BF16, dw=dw16))
        extra = f";  bf16 x: {us16:8.1f} us {fl / us16 / 1e6:6.0f} TFLOP/s, bit-identical: {torch.equal(dw16, dw)}"
        tot += us16 - us
    print(f"{name:16s} {us:8.1f} us  {fl / us / 1e6:7.0f} TFLOP/s   vs exact-f32 kernel: {rel:.2e} relative L2{extra}", flush=True)
    tot += us
print(f"sum {tot:.1f} us (with the bf16-x launches where they exist)")
