"""utilities.video_phasegram (utilities.py:206-228) on the GPU -- the visual input of the phasegram variant
(train_av_net.py:122-125: `y_phasegram = utilities.video_phasegram(y_attn, resize=(p_size, p_size), ...)`)."""
import torch

from . import _lib


def video_phasegram(frames, resize=None, diff=True, cumulative=True, normalize=True):
    """frames [B,1,T,H,W] (cuda) -> phasegram [B,1,T,H*W].  Same arguments as the reference.  The reference's optional
    `resize` goes through torchvision (absent here, and its interpolation is not part of this path): frames must
    already have the phasegram size (32 or 64 square); a `resize` equal to that size is accepted."""
    _lib.require_cuda(frames)
    if frames.dim() != 5 or frames.shape[1] != 1:
        raise ValueError(f"expected attention frames [B,1,T,H,W], got {tuple(frames.shape)}")
    b, _, t, h, w = frames.shape
    if resize is not None and tuple(resize) != (h, w):
        raise NotImplementedError(f"resize {tuple(resize)} != frame size {(h, w)}: resize the attention frames before the call "
                                  "(the reference uses torchvision.transforms.functional.resize, utilities.py:209)")
    if h != w or h not in (32, 64):
        raise ValueError(f"phasegram frames must be 32x32 or 64x64, got {h}x{w}")
    x = frames.contiguous().float()
    out = torch.empty(b, 1, t, h * w, device=x.device, dtype=torch.float32)
    ws = torch.empty(b, t, h * w, device=x.device, dtype=torch.float32)
    amax = torch.empty(1, device=x.device, dtype=torch.float32)
    _lib.call("maavss_video_phasegram", _lib.ptr(x), b, t, h, 1 if diff else 0, 1 if cumulative else 0, 1 if normalize else 0,
              _lib.ptr(ws), _lib.ptr(amax), _lib.ptr(out), _lib.stream_ptr())
    return out
