"""Timing of the Conv3d implicit-GEMM launches (forward + input gradient) of one training step at the benched shape (B=32, T=16, 224^2;
16-bit path: forward operands IEEE half, gradient operands bf16), each checked against the exact-f32 kernel.  GPU box only.
MAAVSS_LIB=<other build> for an A/B."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maavss_amd import _lib
if os.environ.get("MAAVSS_LIB"):
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
# (name, mode of conv3d_prep, C_in of the launch, C_out of the launch, H = W of the input, MFMA mode)
CASES = (("fwd 16->32 @112^2", 0, 16, 32, 112, ops.MODE_F16), ("fwd 32->64 @56^2", 0, 32, 64, 56, ops.MODE_F16),
         ("fwd 64->64 @28^2", 0, 64, 64, 28, ops.MODE_F16), ("dgrad 64->64 @28^2", 1, 64, 64, 28, ops.MODE_BF16),
         ("dgrad 64->32 @56^2", 1, 64, 32, 56, ops.MODE_BF16), ("dgrad 32->16 @112^2", 1, 32, 16, 112, ops.MODE_BF16))
for name, mode, ci, co, hw, pr in CASES:
    dt = torch.float16 if pr == ops.MODE_F16 else torch.bfloat16
    x = (torch.randn(32, 16, hw, hw, ci, device="cuda", generator=g) * 0.5).to(dt)
    w = torch.randn(*((co, ci) if mode == 0 else (ci, co)), 3, 5, 5, device="cuda", generator=g) * 0.05
    wt = ops.conv3d_prep(w, mode, pr)
    us = timed(lambda: ops.conv3d_igemm(x, wt, co, 2, pr, want_stats=(mode == 0)))
    y, part = ops.conv3d_igemm(x, wt, co, 2, pr, want_stats=(mode == 0))
    ref, rpart = ops.conv3d_igemm(x.float(), ops.conv3d_prep(w, mode, ops.MODE_F32), co, 2, ops.MODE_F32, want_stats=(mode == 0))
    rel = ((y - ref).norm() / ref.norm()).item()
    srel = ((part.sum(0) - rpart.sum(0)).norm() / rpart.sum(0).norm()).item() if mode == 0 else 0.0
    fl = 2.0 * 32 * 16 * hw * hw * 75 * ci * co
    print(f"{name:22s} {us:8.1f} us  {fl / us / 1e6:7.0f} TFLOP/s   vs exact-f32 kernel: {rel:.2e} relative L2, BatchNorm sums {srel:.1e}", flush=True)
    tot += us
    del x, y, ref
print(f"sum {tot:.1f} us")
