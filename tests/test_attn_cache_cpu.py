"""Attention-frame cache (storage half): naming, skip-if-present, round trip within the JPEG error."""
import os

import torch

from maavss_amd import attn_cache


def test_save_load_round_trip(tmp_path):
    g = torch.Generator().manual_seed(0)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, 64), torch.linspace(0, 1, 64), indexing="ij")
    clip = torch.stack([(0.5 + 0.5 * torch.sin(6 * xx + t) * torch.cos(4 * yy)) for t in range(5)]).unsqueeze(0)   # [1,5,64,64], smooth
    written = attn_cache.save_frames(clip, str(tmp_path), offset=10)
    assert [os.path.basename(p) for p in written] == [f"img_{i:05d}.jpg" for i in range(10, 15)]
    got = attn_cache.load_cached_frames(str(tmp_path), 10, 5)
    assert got.shape == clip.shape and got.dtype == torch.float32
    assert (got - clip).abs().mean().item() < 0.01 and (got - clip).abs().max().item() < 0.08      # JPEG q75 on a smooth map
    assert attn_cache.load_cached_frames(str(tmp_path), 11, 5) is None                               # img_00015 missing
    # check_exists: present files are left alone
    before = os.path.getmtime(written[0])
    again = attn_cache.save_frames(torch.zeros_like(clip), str(tmp_path), offset=10, check_exists=True)
    assert again == [] and os.path.getmtime(written[0]) == before
    # the dataset's cache-miss branch writes the same layout
    attn_cache.cache_frames(clip, str(tmp_path / "vid"), 3)
    assert attn_cache.verify_files([attn_cache.frame_path(str(tmp_path / "vid"), i) for i in range(3, 8)])
    back = attn_cache.load_cached_frames(str(tmp_path / "vid"), 3, 5)
    assert (back - got).abs().max().item() < 1e-6


def test_grey_weights(tmp_path):
    from PIL import Image
    import numpy as np
    arr = np.zeros((8, 8, 3), np.uint8)
    arr[..., 0], arr[..., 1], arr[..., 2] = 200, 100, 50
    Image.fromarray(arr).save(tmp_path / "img_00000.png")
    os.rename(tmp_path / "img_00000.png", tmp_path / "img_00000.jpg")      # lossless content under the cache's name
    got = attn_cache.load_cached_frames(str(tmp_path), 0, 1)
    want = (0.2989 * 200 + 0.587 * 100 + 0.114 * 50) / 255
    assert abs(got.mean().item() - want) < 1e-6
