"""In-process A/B of the K = 384 ViT GEMMs at the benched shape (M = 512 frames x 785 tokens): panel-stationary
(maavss_vit_panel_gemm, LayerNorm fused) against LayerNorm + weight-stationary (maavss_vit_ws_gemm).
Run on the GPU box:  python3 scripts/gemm_bench.py [--dtype 2] [--frames 512]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maavss_amd import _lib  # noqa: E402
from maavss_amd._lib import call, ptr, stream_ptr  # noqa: E402

if os.environ.get("MAAVSS_LIB"):          # measurement builds (make -C maavss_amd/csrc ablate)
    _lib.LIB_PATH = os.environ["MAAVSS_LIB"]


REPS = 20


def timed(fn, reps=None, warm=3):
    reps = reps or REPS
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", type=int, default=2)
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--ntok", type=int, default=785)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--zeros", action="store_true", help="all-zero operands: the clock give-back test (MI355X_MICROARCH.md, DVFS)")
    args = ap.parse_args()
    global REPS
    REPS = args.reps
    dt = args.dtype
    tdt = torch.float16 if dt == 2 else torch.bfloat16
    m = args.frames * args.ntok
    mp = (m + 127) // 128 * 128
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(mp, 384, device=dev, generator=g)
    xn = torch.randn(mp, 384, device=dev, generator=g).to(tdt)
    if args.zeros:
        x.zero_()
        xn.zero_()
    ln_g, ln_b = torch.ones(384, device=dev), torch.zeros(384, device=dev)
    st = stream_ptr()
    res = {}
    for name, n, epi in (("qkv", 1152, 0), ("fc1", 1536, 1), ("proj", 384, 2)):
        w = (torch.randn(n, 384, device=dev, generator=g) * 384 ** -0.5).to(tdt)
        if args.zeros:
            w.zero_()
        bias = torch.randn(n, device=dev, generator=g) * 0.1
        c = torch.empty(mp, n, device=dev, dtype=torch.float32 if epi == 2 else tdt)
        if epi == 2:
            c.normal_(generator=g)
        xn2 = torch.empty(mp, 384, device=dev, dtype=tdt)
        if epi == 2:
            t_panel = timed(lambda: call("maavss_vit_panel_gemm", None, ptr(xn), 384, None, None, 1e-6, ptr(w), ptr(bias), ptr(c), n, mp,
                                         m, n, epi, 0, 1.0, dt, st))
        else:
            t_panel = timed(lambda: call("maavss_vit_panel_gemm", ptr(x), None, 0, ptr(ln_g), ptr(ln_b), 1e-6, ptr(w), ptr(bias), ptr(c), n, mp,
                                         m, n, epi, 384 if epi == 0 else 0, 0.18, dt, st))
        t_ws = timed(lambda: call("maavss_vit_ws_gemm", ptr(xn), 384, mp, ptr(w), ptr(bias), ptr(c), n, mp, m, n, epi,
                                  384 if epi == 0 else 0, 0.18, None, None, None, 1e-6, dt, st))
        t_ws_ln = None
        if epi == 2:
            t_ws_ln = timed(lambda: call("maavss_vit_ws_gemm", ptr(xn), 384, mp, ptr(w), ptr(bias), ptr(c), n, mp, m, n, epi,
                                         0, 1.0, ptr(xn2), ptr(ln_g), ptr(ln_b), 1e-6, dt, st))
        flops = 2.0 * m * n * 384
        res[name] = (t_panel, t_ws, t_ws_ln)
        print(f"{name:5s} N={n:5d}: panel {t_panel:7.1f} us ({flops / t_panel * 1e-6:6.0f} TFLOP/s) | weight-stationary {t_ws:7.1f} us "
              f"({flops / t_ws * 1e-6:6.0f} TFLOP/s)" + (f" | + LayerNorm out {t_ws_ln:7.1f} us" if t_ws_ln else ""), flush=True)
    t_ln = timed(lambda: call("maavss_vit_layernorm", ptr(x), ptr(ln_g), ptr(ln_b), ptr(xn), m, 384, 1e-6, dt, st))
    print(f"standalone LayerNorm: {t_ln:7.1f} us ({(m * 384 * 6) / t_ln * 1e-6:.2f} TB/s)")
    hid = torch.randn(mp, 1536, device=dev, generator=g).to(tdt)
    w2 = (torch.randn(384, 1536, device=dev, generator=g) * 1536 ** -0.5).to(tdt)
    b2 = torch.zeros(384, device=dev)
    t_fc2 = timed(lambda: call("maavss_vit_gemm", ptr(hid), 1536, ptr(w2), ptr(b2), None, 0, ptr(x), 384, m, 384, 1536, 2, 0, 1.0, dt, st))
    print(f"fc2 (vit_gemm): {t_fc2:7.1f} us")
    now = res["qkv"][0] + res["fc1"][0] + res["proj"][0]
    ws = res["qkv"][1] + res["fc1"][1] + res["proj"][2] + t_ln
    print(f"per block: panel kernels {now:.0f} us | LN + ws(qkv) + ws(proj + LN out) + ws(fc1) {ws:.0f} us")


if __name__ == "__main__":
    main()
