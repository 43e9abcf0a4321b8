"""Timing of the first visual-encoder layer's kernels at the benched shape (B=32, T=16, 224^2): storing form vs recompute passes."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maavss_amd import ops


def timed(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


b, t, h, w = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (32, 16, 224, 224)))
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand(b, t, h, w, device="cuda", generator=g)
wgt = torch.randn(16, 1, 3, 5, 5, device="cuda", generator=g) * 0.1
gamma, beta = 1 + 0.3 * torch.randn(16, device="cuda", generator=g), 0.2 * torch.randn(16, device="cuda", generator=g)
y, part = ops.conv3d_c1_fwd(x, wgt, want_stats=True, precise=ops.MODE_F16)
mean, invstd = ops.bn_finalize(part, b * t * h * w)
out, arg, out16 = ops.bn_pool_act_fwd(y, mean, invstd, gamma, beta, 2, ops.BN_LEAKY, want16=True)
dout = torch.randn(out.shape, device="cuda", generator=g)
dg, db = torch.zeros(16, device="cuda"), torch.zeros(16, device="cuda")
coef = ops.bn_pool_act_bwd(dout, out, arg, y, mean, invstd, gamma, 2, ops.BN_LEAKY, dgamma=dg, dbeta=db, beta=beta, coef_only=True)
dw = torch.empty(16, 1, 3, 5, 5, device="cuda")
res = {
    "c1_fwd (store y + stats)": timed(lambda: ops.conv3d_c1_fwd(x, wgt, want_stats=True, precise=ops.MODE_F16)),
    "bn_pool_act_fwd (reads y)": timed(lambda: ops.bn_pool_act_fwd(y, mean, invstd, gamma, beta, 2, ops.BN_LEAKY, want16=True)),
    "c1_wgrad_bn (reads y)": timed(lambda: ops.conv3d_c1_wgrad_bn(x, y, dout, out, arg, mean, invstd, coef, 2, dw=dw, precise=ops.MODE_BF16)),
    "c1_stats": timed(lambda: ops.conv3d_c1_stats(x, wgt, gamma)),
    "c1_bn_pool_act": timed(lambda: ops.conv3d_c1_bn_pool_act(x, wgt, mean, invstd, gamma, beta)),
    "c1_wgrad_bn_recompute": timed(lambda: ops.conv3d_c1_wgrad_bn_recompute(x, wgt, dout, arg, mean, invstd, beta, coef, 2, dw=dw)),
}
for k, v in res.items():
    print(f"{k:28s} {v:8.1f} us", flush=True)
v = list(res.values())
print(f"storing path {sum(v[:3]):.1f} us, recompute path {sum(v[3:]):.1f} us")
