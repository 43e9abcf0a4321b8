"""VideoAttention.__load_model on a synthetic DINO-style checkpoint (video_attention.py:116-129: key "teacher",
prefixes "module." / "backbone." stripped, strict=False) -- CPU only, no kernel runs; plus the reference's treatment
of `resize` (stored, never read: video_attention.py:29-30)."""
import torch

import maavss_amd
from maavss_amd.video_attention import interpolate_pos_embed, vit_small_shapes
from oracle import vit_ref_cpu as vref


def _dino_style_checkpoint(path, img_size=224, seed=3):
    sd = vref.seeded_vit_state(seed, img_size)
    teacher = {"module.backbone." + k: v for k, v in sd.items()}
    teacher["module.head.mlp.0.weight"] = torch.zeros(4, 4)          # DINO head entries the ViT does not have
    torch.save({"student": {"module.backbone.cls_token": torch.zeros(1, 1, 384)}, "teacher": teacher, "epoch": 7}, path)
    return sd


def test_loads_teacher_key_and_strips_prefixes(tmp_path):
    path = str(tmp_path / "dino_deitsmall8_pretrain.pth")
    sd = _dino_style_checkpoint(path)
    va = maavss_amd.VideoAttention(path_to_weights=path, device="cpu")
    assert va.model.loaded_from == path
    got = va.model.state_dict()
    assert set(got) == set(vit_small_shapes())
    for k, v in sd.items():
        assert torch.equal(got[k], v), k


def test_bare_state_dict_without_teacher_key(tmp_path):
    path = str(tmp_path / "bare.pth")
    sd = vref.seeded_vit_state(5)
    torch.save({"backbone." + k: v for k, v in sd.items()}, path)
    va = maavss_amd.VideoAttention(path_to_weights=path, device="cpu")
    assert all(torch.equal(va.model.state_dict()[k], v) for k, v in sd.items())


def test_non_224_position_embedding_is_kept_and_interpolated(tmp_path):
    path = str(tmp_path / "vit384.pth")
    sd = _dino_style_checkpoint(path, img_size=384)                   # pos_embed [1, 2305, 384]
    va = maavss_amd.VideoAttention(path_to_weights=path, device="cpu")
    assert tuple(va.model.sd["pos_embed"].shape) == (1, 2305, 384)
    pe = interpolate_pos_embed(va.model.sd["pos_embed"], 28, 28)      # a 224^2 frame through 384^2 weights
    assert tuple(pe.shape) == (1, 785, 384)
    assert torch.equal(pe, vref.interpolate_pos_embed(sd["pos_embed"], 28, 28))
    assert torch.equal(pe[:, 0], sd["pos_embed"][:, 0])               # the CLS position is never resized


def test_missing_file_keeps_seeded_init_and_resize_is_ignored(capsys):
    va = maavss_amd.VideoAttention(path_to_weights="/nonexistent/dino.pth", resize=(112, 112), device="cpu")
    assert va.model.loaded_from is None and va.resize == (112, 112)
    assert "not found" in capsys.readouterr().err          # a notice, on stderr: bench.py's stdout is exactly one JSON line


def test_video_attention_mode_arguments_are_validated_on_the_host():
    """Round 4 options of the drop-in extractor (none of them in the reference's signature): the block-range hybrid of the fp8 attention and
    the packed-half GELU.  Construction touches no device."""
    import pytest
    from maavss_amd.video_attention import DEPTH, VideoAttention
    kw = dict(path_to_weights="/nonexistent.pth")
    assert VideoAttention(**kw).fp8_blocks == frozenset() and VideoAttention(**kw).gelu_epilogue == 1
    assert VideoAttention(attn_dtype="fp8", **kw).fp8_blocks == frozenset(range(DEPTH - 1))
    assert VideoAttention(attn_dtype="fp8-late", **kw).fp8_blocks == frozenset((8, 9, 10))
    assert VideoAttention(attn_dtype="fp8", fp8_blocks=(9, 10), **kw).fp8_blocks == frozenset((9, 10))
    assert VideoAttention(gelu="half", **kw).gelu_epilogue == 4
    assert VideoAttention(**kw).qkv_ln == "pre" and VideoAttention(qkv_ln="post", **kw).qkv_ln == "post"
    for bad in (dict(attn_dtype="fp4"), dict(fp8_blocks=(3,)), dict(attn_dtype="fp8", fp8_blocks=(11,)), dict(gelu="half", act_dtype="bf16"),
                dict(gelu="tanh"), dict(act_dtype="fp32"), dict(patch_size=16), dict(qkv_ln="both")):
        with pytest.raises(ValueError):
            VideoAttention(**kw, **bad)
