"""Where the cycles of a vit_ws_gemm wave go (measurement build: hipcc -DWS_STAMP, lib/libmaavss_wsstamp.so; s_memtime at the phase
boundaries of the panel loop, summed per wave).  GPU box:  python scripts/ws_stamp.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maavss_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libmaavss_wsstamp.so")
from maavss_amd._lib import call, ptr, stream_ptr  # noqa: E402

L = _lib.lib()
rd = L.cdll.maavss_ws_stamps_read
rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
m = 512 * 785
mp = (m + 127) // 128 * 128
dev, dt, tdt = "cuda", 2, torch.float16
g = torch.Generator(device=dev).manual_seed(1)
xn = torch.randn(mp, 384, device=dev, generator=g).to(tdt)
x = torch.randn(mp, 384, device=dev, generator=g)
stats = torch.stack([x.view(mp, 3, 128).mean(-1), ((x.view(mp, 3, 128) - x.view(mp, 3, 128).mean(-1, keepdim=True)) ** 2).sum(-1)], -1).contiguous()
ln_g, ln_b = torch.ones(384, device=dev), torch.zeros(384, device=dev)
st = stream_ptr()
names = ["barrier wait", "frag reads + MFMAs", "epilogue", "fetch wait", "deposit"]
for name, n, epi, ln in (("qkv (norm1 on the way in)", 1152, 0, True), ("qkv (16-bit input)", 1152, 0, False), ("fc1", 1536, 1, False), ("proj + norm2", 384, 2, False)):
    w = (torch.randn(n, 384, device=dev, generator=g) * 384 ** -0.5).to(tdt)
    bias = torch.randn(n, device=dev, generator=g) * 0.1
    c = torch.empty(mp, n, device=dev, dtype=torch.float32 if epi == 2 else tdt)
    xn2 = torch.empty(mp, 384, device=dev, dtype=tdt)
    if ln:
        f = lambda: call("maavss_vit_ws_gemm_ln", ptr(x), mp, ptr(stats), ptr(ln_g), ptr(ln_b), 1e-6, ptr(w), ptr(bias), ptr(c), n, mp, m, n, 384, 0.18, dt, st)
    elif epi == 2:
        c.normal_(generator=g)
        f = lambda: call("maavss_vit_ws_gemm", ptr(xn), 384, mp, ptr(w), ptr(bias), ptr(c), n, mp, m, n, 2, 0, 1.0, ptr(xn2), ptr(ln_g), ptr(ln_b), 1e-6, dt, st)
    else:
        f = lambda: call("maavss_vit_ws_gemm", ptr(xn), 384, mp, ptr(w), ptr(bias), ptr(c), n, mp, m, n, epi, 384 if epi == 0 else 0, 0.18, None, None, None, 1e-6, dt, st)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    buf = np.zeros(256 * 12 * 8, dtype=np.uint64)
    assert rd(buf.ctypes.data, 1) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    f()
    e1.record()
    torch.cuda.synchronize()
    assert rd(buf.ctypes.data, 1) == 0
    b = buf.reshape(256, 12, 8).astype(np.float64)
    live = b[:, :, 5] > 0
    panels = b[:, :, 5][live]
    per = b[:, :, :5][live] / panels[:, None]          # ticks per panel and wave
    tot = per.sum(1)
    print(f"{name}: {e0.elapsed_time(e1) * 1e3:.0f} us with stamps; {panels.mean():.1f} panels per wave; s_memtime ticks per PANEL and wave (median over {per.shape[0]} waves): "
          + ", ".join(f"{nm} {np.median(per[:, i]):.0f}" for i, nm in enumerate(names)) + f"; sum {np.median(tot):.0f}", flush=True)
