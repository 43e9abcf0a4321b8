import sys, torch
sys.path.insert(0, '.')
from maavss_amd import _lib
def case(ntok, frames, ramp, dt):
    tdt = {0: torch.bfloat16, 2: torch.float16}[dt]
    rows = frames * ntok
    qkv = torch.randn(rows, 1152, generator=torch.Generator().manual_seed(5))
    qkv[:, :384] *= 0.125 * 3 * 1.4426950408889634
    if ramp:
        scale = torch.linspace(1.0, abs(ramp), ntok).repeat(frames)
        if ramp < 0: scale = scale.flip(0)
        qkv[:, 384:768] = qkv[:, 384:768] * scale[:, None]
    qkv = qkv.to(tdt)
    out = torch.empty(rows, 384, dtype=tdt, device="cuda")
    qc = qkv.cuda()
    _lib.call("maavss_vit_attn", qc.data_ptr(), out.data_ptr(), frames, ntok, 6, 1152, 384, dt, _lib.stream_ptr())
    got = out.float().cpu()
    bad = ~torch.isfinite(got)
    q, k, v = [t.view(frames, ntok, 6, 64).transpose(1, 2) for t in qkv.double().split(384, 1)]
    s = q @ k.transpose(-1, -2)
    print(f"ntok {ntok} ramp {ramp} dt {dt}: nonfinite {int(bad.sum())} rows {sorted(set(bad.nonzero()[:,0].tolist()))[:20]} cols {sorted(set(bad.nonzero()[:,1].tolist()))[:10]}; score range {s.min().item():.1f} {s.max().item():.1f}")
    if bad.any():
        r = bad.nonzero()[0,0].item(); c = bad.nonzero()[0,1].item()
        hd = c // 64
        srow = s[r // ntok, hd, r % ntok]
        tm = [srow[t:t+64].max().item() for t in range(0, ntok, 64)]
        print("   first bad row", r, "head", hd, "tile maxima", [round(x,1) for x in tm])
for dt in (2, 0):
    case(785, 1, 6.0, dt); case(300, 2, -6.0, dt); case(129, 1, 3.0, dt)
