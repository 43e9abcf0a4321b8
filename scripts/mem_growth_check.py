"""300 steps of the two-stream training loop (extraction on the side stream, TrainStep on the main one) at B = 8: the device allocator's
allocated / reserved MiB after steps 20, 100 and 300 must not grow (events, pinned flag copies and slots are recycled)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import maavss_amd
b, t, w, hpf, fft = 8, 16, 224, 8, 512
hop, length, t_a = maavss_amd.calc_hop_size(t, hpf, 30, 16000)
dev = torch.device('cuda:0')
va = maavss_amd.VideoAttention(path_to_weights='/nonexistent.pth')
st = maavss_amd.STFT(fft, hop, noise_std=0.1, device=dev)
model = maavss_amd.AV_Fusion_Model_Frames([b, 2, hpf * t, fft // 2 + 1], [b, 1, t, w, w], hpf, spatial_match='adaptive').to(dev).train()
step = maavss_amd.TrainStep(model, lr=1e-5)
pipe = maavss_amd.ClipPipeline(va, st, clip_frames=t)
g = torch.Generator(device='cuda').manual_seed(0)
def batch():
    return torch.rand(b * t, 3, w, w, device=dev, generator=g), (0.3 * torch.randn(b, length, device=dev, generator=g)).clamp(-1, 1)
pipe.submit(*batch(), seed=0)
marks = {}
for i in range(301):
    pipe.submit(*batch(), seed=i + 1)
    attn, x, y = pipe.get()
    mid = t // 2
    step(x, attn, y[:, :, mid * hpf:(mid + 1) * hpf].contiguous(), attn[:, :, mid].contiguous())
    pipe.release()
    if i in (20, 100, 300):
        torch.cuda.synchronize()
        marks[i] = (torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20)
pipe.drain()
print(marks)
assert marks[300][1] <= marks[20][1] * 1.02 + 64, marks
print("no growth of the device allocator between step 20 and step 300")
