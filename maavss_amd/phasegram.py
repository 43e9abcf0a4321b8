"""utilities.video_phasegram (utilities.py:206-228) on the GPU -- the visual input of the phasegram variant
(train_av_net.py:122-125: `y_phasegram = utilities.video_phasegram(y_attn, resize=(p_size, p_size), ...)`)."""
import torch

from . import _lib


def video_phasegram(frames, resize=None, diff=True, cumulative=True, normalize=True):
    """frames [B,1,T,H,W] (cuda) -> phasegram [B,1,T,h*w].  Same arguments as the reference.  `resize=(h, w)` is the
    reference's torchvision resize of a tensor, i.e. bilinear interpolation with align_corners=False and no antialiasing
    (K20's resize kernel); the phasegram itself needs 32x32 or 64x64 frames."""
    _lib.require_cuda(frames)
    if frames.dim() != 5 or frames.shape[1] != 1:
        raise ValueError(f"expected attention frames [B,1,T,H,W], got {tuple(frames.shape)}")
    b, _, t, h, w = frames.shape
    x = frames.contiguous().float()
    if resize is not None and tuple(resize) != (h, w):
        rh, rw = int(resize[0]), int(resize[1])
        small = torch.empty(b, 1, t, rh, rw, device=x.device, dtype=torch.float32)
        _lib.call("maavss_resize_bilinear", _lib.ptr(x), _lib.ptr(small), b * t, h, w, rh, rw, _lib.stream_ptr())
        x, h, w = small, rh, rw
    if h != w or h not in (32, 64):
        raise ValueError(f"phasegram frames must be 32x32 or 64x64 (after resize), got {h}x{w}")
    out = torch.empty(b, 1, t, h * w, device=x.device, dtype=torch.float32)
    ws = torch.empty(b, t, h * w, device=x.device, dtype=torch.float32)
    amax = torch.empty(1, device=x.device, dtype=torch.float32)
    _lib.call("maavss_video_phasegram", _lib.ptr(x), b, t, h, 1 if diff else 0, 1 if cumulative else 0, 1 if normalize else 0,
              _lib.ptr(ws), _lib.ptr(amax), _lib.ptr(out), _lib.stream_ptr())
    return out
