"""On-disk cache of attention frames (the storage half): one directory per video, one JPEG per frame,
`img_%05d.jpg`, written by save_attn_videos.save_frames (save_attn_videos.py:10-19) and by AV_Dataset.get_frames_cached
(av_dataset.py:251-278), read back through `ToTensor() -> Grayscale()` (av_dataset.py:84-87, 264-266).

File / codec work, not GPU work: tensors are taken from and returned on the CPU (or the caller's device for `load`).
torchvision is absent here; its two helpers on this path are restated on PIL, which is what they call themselves:
  torchvision.utils.save_image(x, path)  = Image.fromarray((x * 255 + 0.5).clamp(0, 255).uint8, HWC, grey -> 3 channels)
                                           .save(path)                       (JPEG by extension, PIL defaults)
  ToTensor() -> Grayscale()              = uint8 HWC / 255 -> 0.2989 R + 0.587 G + 0.114 B
Parity unpinned: JPEG is lossy and no fixture of the reference's cache exists; tests check the round trip within the
codec's error and the file naming / skip-if-present behaviour.
"""
import os

import numpy as np
import torch
from PIL import Image


def frame_path(folder, index):
    return os.path.join(folder, f"img_{index:05d}.jpg")


def _save_image(frame, path):
    """frame [H,W], [1,H,W] or [3,H,W] float in [0,1] -> 8-bit RGB file (torchvision.utils.save_image semantics)."""
    f = frame.detach().to("cpu", torch.float32)
    if f.dim() == 2:
        f = f.unsqueeze(0)
    if f.shape[0] == 1:
        f = f.repeat(3, 1, 1)
    arr = f.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    Image.fromarray(arr).save(path)


def save_frames(frames, save_path, offset=0, check_exists=False):
    """save_attn_videos.save_frames: frames [1,T,H,W] (or [T,H,W]) -> save_path/img_{i+offset:05d}.jpg."""
    attn = frames.squeeze(0) if frames.dim() == 4 else frames
    written = []
    for i, frame in enumerate(attn):
        path = frame_path(save_path, i + offset)
        if check_exists and os.path.isfile(path):
            print(f"{path} already exists, skipping...")
            continue
        _save_image(frame, path)
        written.append(path)
    return written


def verify_files(files):
    """utilities.verify_files (utilities.py:419-423)."""
    return all(os.path.isfile(f) for f in files)


def _load_grey(path):
    arr = np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)
    t = torch.from_numpy(arr.copy()).permute(2, 0, 1).to(torch.float32).div_(255)          # ToTensor
    return (0.2989 * t[0] + 0.587 * t[1] + 0.114 * t[2]).unsqueeze(0)                         # Grayscale (1 channel)


def load_cached_frames(folder_path, true_idx, num_frames, device="cpu"):
    """The cache-hit branch of AV_Dataset.get_frames_cached (av_dataset.py:251-268): frames true_idx .. +num_frames of one
    video -> attention clip [1, T, H, W], or None when any file is missing (the caller then generates and caches)."""
    paths = [frame_path(folder_path, i + true_idx) for i in range(num_frames)]
    if not verify_files(paths):
        return None
    frames = [_load_grey(p) for p in paths]
    return torch.stack(frames, dim=1).to(device)


def cache_frames(attn, folder_path, true_idx):
    """The cache-miss branch (av_dataset.py:273-278): attn [1, T, H, W] -> one 3-channel JPEG per frame."""
    os.makedirs(folder_path, exist_ok=True)
    for i in range(attn.shape[1]):
        _save_image(attn[:, i, :, :], frame_path(folder_path, i + true_idx))
