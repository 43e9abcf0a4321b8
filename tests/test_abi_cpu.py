"""CPU-only: the C-ABI library loads and exports every symbol include/maavss.h declares."""
import ctypes
import os

import pytest

from maavss_amd import _lib


def test_library_exports_every_declared_symbol():
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    protos = _lib.parse_header()
    assert len(protos) >= 5
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(cdll, name), f"{name} declared in include/maavss.h but not exported"
    L = _lib.lib()
    assert L.cdll.maavss_arch() == b"gfx950"
    assert L.cdll.maavss_version() == _lib.header_abi_version() == 400


def test_no_cpu_fallback():
    import torch
    import maavss_amd
    st = maavss_amd.STFT(512, 66, device="cpu")
    with pytest.raises(_lib.MaavssError):
        st(torch.zeros(1, 4224))


def test_round3_entry_points_validate_their_arguments_before_touching_the_device():
    """Host-side contract of the entry points added in round 3 (no compute: every call below fails its argument check, or is a pure
    size / switch query): status codes, messages, workspace sizes as laid out in maavss_amd/csrc/vit_mx.h."""
    L = _lib.lib()
    # block-scaled fp8 attention workspace: q8 + k8 + v8t (rows_alloc * 384 each) + q / k scales (12 planes each) + v scales + 256
    for rows in (1, 785, 401920):
        ra = (rows + 127) // 128 * 128 + 128
        assert _lib.query("maavss_vit_attn_mx_ws_bytes", rows) == 3 * ra * 384 + 2 * 12 * ra + 384 * (ra // 32) + 256
    assert _lib.query("maavss_vit_attn_mx_ws_bytes", 0) == 0
    with pytest.raises(_lib.MaavssError, match="vit_attn_mx"):
        _lib.call("maavss_vit_attn_mx", None, None, 1, 785, 6, 384, 2, None)
    with pytest.raises(_lib.MaavssError, match="6 heads"):
        _lib.call("maavss_vit_attn_mx", 256, 256, 1, 785, 8, 512, 2, None)          # only ViT-S (6 x 64) is built
    with pytest.raises(_lib.MaavssError, match="dtype"):
        _lib.call("maavss_vit_attn_mx", 256, 256, 1, 785, 6, 384, 1, None)
    with pytest.raises(_lib.MaavssError, match="256-byte aligned"):
        _lib.call("maavss_vit_attn_mx", 264, 256, 1, 785, 6, 384, 2, None)
    with pytest.raises(_lib.MaavssError, match="vit_qkv_mx"):
        _lib.call("maavss_vit_qkv_mx", 256, 256, 0, 1152, 2, None)
    with pytest.raises(_lib.MaavssError, match="vit_ws_gemm_ln_mx"):
        _lib.call("maavss_vit_ws_gemm_ln_mx", None, 128, None, None, None, 1e-6, None, None, None, 100, 384, 1.0, 2, None)
    with pytest.raises(_lib.MaavssError, match="multiple of 8"):
        _lib.call("maavss_f32_to_bf16", 256, 512, 12, None)
    with pytest.raises(_lib.MaavssError, match="16-byte aligned"):
        _lib.call("maavss_bf16_to_f32", 260, 512, 16, None)
    # first-layer forward: one BatchNorm partial per workgroup of 8 tiles (MFMA form), per tile (exact-f32 form)
    tiles = 14 * 14 * 32 * 16
    assert _lib.query("maavss_conv3d_c1_fwd_nparts", 32, 16, 224, 224, 2) == (tiles + 7) // 8
    assert _lib.query("maavss_conv3d_c1_fwd_nparts", 32, 16, 224, 224, 1) == tiles
    # the first layer without its conv output: null operands and oversize problems are refused before any launch
    with pytest.raises(_lib.MaavssError, match="conv3d_c1_stats"):
        _lib.call("maavss_conv3d_c1_stats", 256, 256, None, 256, 256, 1, 1, 16, 16, None)
    with pytest.raises(_lib.MaavssError, match="too many tiles"):
        _lib.call("maavss_conv3d_c1_stats", 256, 256, 256, 256, 256, 1 << 15, 1 << 15, 1024, 1024, None)
    with pytest.raises(_lib.MaavssError, match="conv3d_c1_bn_pool_act"):
        _lib.call("maavss_conv3d_c1_bn_pool_act", 256, 256, 256, 256, 256, 256, 256, None, None, None, 1, 1, 16, 16, None)   # argmax is required
    with pytest.raises(_lib.MaavssError, match="pool must be 2 or 3"):
        _lib.call("maavss_conv3d_c1_wgrad_bn_recompute", 256, 256, 256, 256, 256, 256, 256, 256, 4, 256, 256, 8, 1, 1, 16, 16, 0, None)
    with pytest.raises(_lib.MaavssError, match="null pointer"):
        _lib.call("maavss_conv3d_c1_wgrad_bn_recompute", 256, 256, 256, 256, 256, 256, None, 256, 2, 256, 256, 8, 1, 1, 16, 16, 0, None)   # BatchNorm bias
    # deterministic switch: process-wide, returns the previous setting
    prev = _lib.query("maavss_set_deterministic", 1)
    assert _lib.query("maavss_get_deterministic") == 1
    assert _lib.query("maavss_set_deterministic", prev) == 1
    assert _lib.query("maavss_get_deterministic") == prev
    with pytest.raises(_lib.MaavssError, match="aligned"):
        _lib.call("maavss_set_deterministic_workspace", 260, 1024, None)
    _lib.call("maavss_set_deterministic_workspace", None, 0, None)


def test_round4_argument_checks():
    """the first-layer weight-gradient entry points refuse images smaller than the pool window / beyond the kernel's
    index arithmetic, the fp8 attention refuses workspaces beyond its 32-bit offsets -- all before any launch."""
    with pytest.raises(_lib.MaavssError, match=r"H, W must be in \[pool, 98304\)"):
        _lib.call("maavss_conv3d_c1_wgrad_bn_recompute", 256, 256, 256, 256, 256, 256, 256, 256, 3, 256, 256, 8, 1, 1, 2, 16, 0, None)      # H < pool
    with pytest.raises(_lib.MaavssError, match=r"H, W must be in \[pool, 98304\)"):
        _lib.call("maavss_conv3d_c1_wgrad_bn_recompute", 256, 256, 256, 256, 256, 256, 256, 256, 3, 256, 256, 8, 1, 1, 16, 98304, 0, None)
    with pytest.raises(_lib.MaavssError, match="too many tiles"):
        _lib.call("maavss_conv3d_c1_wgrad_bn_recompute", 256, 256, 256, 256, 256, 256, 256, 256, 2, 256, 256, 8, 1 << 12, 1 << 12, 4096, 4096, 0, None)
    with pytest.raises(_lib.MaavssError, match="H, W must be in"):
        _lib.call("maavss_conv3d_c1_wgrad_bn", 256, 256, 256, 256, 256, 256, 256, 256, 2, 256, 256, 8, 1, 1, 1, 16, 0, 0, None)
    with pytest.raises(_lib.MaavssError, match="empty image or too many tiles"):
        _lib.call("maavss_conv3d_c1_wgrad", 256, 256, 256, 256, 8, 1 << 12, 1 << 12, 4096, 4096, 0, None)
    # 4 608 frames x 785 tokens: 3 * rows_alloc * 384 > 2^32
    rows = 4608 * 785
    assert _lib.query("maavss_vit_attn_mx_ws_bytes", rows) >= 1 << 32
    with pytest.raises(_lib.MaavssError, match="32-bit"):
        _lib.call("maavss_vit_attn_mx", 256, 256, 4608, 785, 6, 384, 2, None)
    with pytest.raises(_lib.MaavssError, match="32-bit"):
        _lib.call("maavss_vit_qkv_mx", 256, 256, rows, 1152, 2, None)
    with pytest.raises(_lib.MaavssError, match="32-bit"):
        _lib.call("maavss_vit_ws_gemm_ln_mx", 256, (rows + 63) // 64 * 64, 256, 256, 256, 1e-6, 256, 256, 256, rows, 384, 1.0, 2, None)
