"""Per-stage distance of the HIP forward pass to the CPU oracle twin (fp32 and 16-bit-emulating), to find which stage
carries the difference that is not operand rounding.  GPU box only; oracle = checker."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import maavss_amd
from oracle import avse_ref_cpu as orc

def stages_twin(twin, x_a, x_v):
    outs = {}
    hooks = []
    for i in range(5):
        hooks.append(twin.visual_encoder[4 * i].register_forward_hook(lambda m, a, o, i=i: outs.__setitem__(f"vis{i}.y", o.detach())))
        hooks.append(twin.visual_encoder[4 * i + 3].register_forward_hook(lambda m, a, o, i=i: outs.__setitem__(f"vis{i}.out", o.detach())))
    hooks.append(twin.stft_encoder.register_forward_hook(lambda m, a, o: outs.__setitem__("aud.out", o.detach())))
    hooks.append(twin.lstm.register_forward_hook(lambda m, a, o: outs.__setitem__("lstm", o[0].detach())))
    hooks.append(twin.fc1.register_forward_hook(lambda m, a, o: outs.__setitem__("h1", torch.tanh(o.detach()))))
    a, v, f = twin(x_a, x_v)
    outs.update(fused=f.detach(), a=a.detach(), v=v.detach())
    for h in hooks:
        h.remove()
    return outs

def stages_hip(model, x_a, x_v):
    (a, v, f), sv = model._engine_forward(x_a.cuda(), x_v.cuda(), train=True)
    b, t = x_v.shape[0], model.t_v
    outs = {}
    for i in range(5):
        y = sv["vis"][i]["y"]                       # [B,T,H,W,C] channels-last
        outs[f"vis{i}.y"] = y.permute(0, 4, 1, 2, 3).float().cpu()
        if i < 4:
            outs[f"vis{i}.out"] = sv["vis"][i]["out"].permute(0, 4, 1, 2, 3).float().cpu()
    ts = t * model.s_v
    seq = sv["seq"].cpu()
    outs["vis4.out"] = seq[:, :, :ts].reshape(b, 16, t, model.side, model.side)
    outs["aud.out"] = seq[:, :, ts:].reshape(b, 16, t, model.s_v)
    outs["lstm"] = sv["av"].cpu().view(b, 16, 512)
    outs["h1"] = sv["h1"].cpu()
    outs.update(fused=f.cpu(), a=a.cpu(), v=v.cpu())
    return outs

def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "P"
    b, t, w, sp = (2, 8, 256, "exact") if which == "P" else (2, 16, 224, "adaptive")
    hpf, fft = 8, 512
    shapes = ([b, 2, hpf * t, fft // 2 + 1], [b, 1, t, w, w], hpf)
    x_a, x_v, y_a, y_v = orc.synthetic_batch(b, t, w, hpf * t, fft // 2 + 1, hpf, 42)
    for precise in (True, False):
        twin = orc.AVFusionFramesRef(*shapes, spatial_match=sp, emulate_16bit=not precise)
        model = maavss_amd.AV_Fusion_Model_Frames(*shapes, precise=precise, spatial_match=sp)
        model.load_state_dict(orc.seeded_state_dict(twin, 41)); orc.load_seeded(twin, 41)
        model = model.cuda().train(); twin.train()
        with torch.no_grad():
            ref = stages_twin(twin, x_a, x_v)
            got = stages_hip(model, x_a, x_v)
        print(f"--- precise={precise}: HIP vs {'fp32' if precise else '16-bit-emulating'} twin, shape {which}")
        for k in ref:
            r, g = ref[k].float(), got[k].reshape(ref[k].shape).float()
            d = g - r
            print(f"{k:10s} rms(ref) {r.pow(2).mean().sqrt().item():.3e}  rms(diff) {d.pow(2).mean().sqrt().item():.3e}  max|diff| {d.abs().max().item():.3e}  rel {d.norm().item() / (r.norm().item() + 1e-30):.3e}")
main()
