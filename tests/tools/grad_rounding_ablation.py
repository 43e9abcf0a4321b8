"""Which rounding of the 16-bit conv path moves the gradients, and by how much (CPU, oracle twin).

    python tests/tools/grad_rounding_ablation.py P        > profiles/r3_grad_rounding_ablation.txt
    python tests/tools/grad_rounding_ablation.py benched >> profiles/r3_grad_rounding_ablation.txt

Every Conv3d of the oracle twin (oracle/avse_ref_cpu.py) is replaced by an autograd node whose operand roundings are
switchable one by one: forward x / w per layer, input-gradient dy / w, weight-gradient x / dy (layer 0, layer 1, layers 2-4).
Formats: f32 (none), f16 (11 bits), bf16 (8 bits), hilo = bf16 hi + bf16 lo (16 bits; also stands for any two-pass scheme).
Printed: relative L2 distance of gradient tensors to the all-fp32 run, and the mask-MSE of the forward output.
Test infrastructure; uses the oracle, never the product."""
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import avse_ref_cpu as orc  # noqa: E402

bf = lambda t: t.to(torch.bfloat16).float()          # noqa: E731
hf = lambda t: t.half().float()                      # noqa: E731


def hilo(t):
    h = bf(t)
    return h + bf(t - h)


R = {"f32": lambda t: t, "bf16": bf, "f16": hf, "hilo": hilo}
CFG = {}


class Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, pad, layer):
        ctx.save_for_backward(x, w)
        ctx.pad, ctx.layer = pad, layer
        r = R[CFG[f"f{layer}"]]
        return F.conv3d(r(x), r(w), padding=(1, pad, pad))

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        c, layer, pad = CFG, ctx.layer, (1, ctx.pad, ctx.pad)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.nn.grad.conv3d_input(x.shape, R[c["dg_w"]](w), R[c["dg_dy"]](dy), padding=pad)
        key = "wg0" if layer == 0 else ("wg1" if layer == 1 else "wg")
        dw = torch.nn.grad.conv3d_weight(R[c[key + "_x"]](x), w.shape, R[c[key + "_dy"]](dy), padding=pad)
        return dx, dw, None, None


class ConvM(nn.Conv3d):
    layer = 0

    def forward(self, x):
        return Conv.apply(x, self.weight, self.padding[1], self.layer)


def run(cfg, shapes, spatial, batch, seed=41):
    CFG.clear()
    CFG.update(cfg)
    m = orc.AVFusionFramesRef(*shapes, spatial_match=spatial)
    convs = [mod for mod in m.visual_encoder if isinstance(mod, nn.Conv3d)]
    for i, mod in enumerate(convs):
        mod.__class__ = ConvM
        mod.layer = i
    orc.load_seeded(m, seed)
    m.train()
    loss, _, _, (a, _, _) = orc.loss_ref(m, *batch, 0.001, 1)
    loss.backward()
    return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}, a.detach()


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "P"
    b, t, w, sp = (2, 8, 256, "exact") if which == "P" else (2, 16, 224, "adaptive")
    hpf, fft = 8, 512
    shapes = ([b, 2, hpf * t, fft // 2 + 1], [b, 1, t, w, w], hpf)
    batch = orc.synthetic_batch(b, t, w, hpf * t, fft // 2 + 1, hpf, 42)
    torch.set_num_threads(os.cpu_count() or 1)
    base = dict(dg_w="f32", dg_dy="f32", wg_x="f32", wg_dy="f32", wg0_x="f32", wg0_dy="f32", wg1_x="f32", wg1_dy="f32",
                f0="f32", f1="f32", f2="f32", f3="f32", f4="f32")
    fwd = lambda fmt, layers=range(5): {f"f{i}": fmt for i in layers}          # noqa: E731
    bwd16 = dict(dg_w="bf16", dg_dy="bf16", wg_x="bf16", wg_dy="bf16", wg0_x="bf16", wg0_dy="bf16", wg1_x="bf16", wg1_dy="bf16")
    cur = dict(base, **fwd("f16"), **bwd16)
    variants = [
        ("shipped 16-bit path (fwd f16, bwd bf16)", cur),
        ("forward f16 only, backward exact", dict(base, **fwd("f16"))),
        ("input-gradient operands bf16 only", dict(base, dg_w="bf16", dg_dy="bf16")),
        ("weight-gradient operands bf16 only", dict(base, wg_x="bf16", wg_dy="bf16", wg0_x="bf16", wg0_dy="bf16", wg1_x="bf16", wg1_dy="bf16")),
        ("whole backward bf16, forward exact", dict(base, **bwd16)),
        ("shipped + dy hi+lo in wgrad of layers 0,1 (VERDICT r2 item 2)", dict(cur, wg0_dy="hilo", wg1_dy="hilo")),
        ("shipped + that + x as f16 there", dict(cur, wg0_dy="hilo", wg1_dy="hilo", wg0_x="f16", wg1_x="f16")),
        ("forward f16 in layer 0 only", dict(base, f0="f16")),
        ("forward f16 in layer 1 only", dict(base, f1="f16")),
        ("forward f16 in layer 2 only", dict(base, f2="f16")),
        ("forward f16 in layer 3 only", dict(base, f3="f16")),
        ("forward f16 in layer 4 only", dict(base, f4="f16")),
        ("forward bf16 (8 bits) everywhere", dict(base, **fwd("bf16"))),
        ("forward 16 mantissa bits everywhere (two-pass)", dict(base, **fwd("hilo"))),
        ("forward 16 bits in layers 0,1 / f16 in 2-4", dict(base, **fwd("f16", (2, 3, 4)), f0="hilo", f1="hilo")),
    ]
    t0 = time.time()
    ref, a_ref = run(base, shapes, sp, batch)
    print(f"# shape {which}: B={b} T={t} {w}x{w} fft {fft} ({sp}); fp32 reference run {time.time() - t0:.1f} s; relative L2 of gradient tensors to it")
    print(f"# {'variant':62s} mask-MSE   conv0.w    conv1.w    conv2.w    conv3.w    conv4.w    worst tensor")
    for name, cfg in variants:
        g, a = run(cfg, shapes, sp, batch)
        e = {k: ((g[k] - ref[k]).norm() / (ref[k].norm() + 1e-30)).item() for k in ref}
        wk = max(e, key=e.get)
        cols = " ".join(f"{e[f'visual_encoder.{4 * i}.weight']:.3e}" for i in range(5))
        print(f"{name:64s} {((a - a_ref) ** 2).mean().item():.2e}  {cols}  {wk} {e[wk]:.3e}", flush=True)


main()
