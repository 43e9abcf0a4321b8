"""Which 16-bit storage point of the extractor moves the attention frames (and so the mask) most?  CPU only, oracle only.

Runs oracle/vit_ref_cpu.py's rounding-emulating ViT with the rounding restricted to (a) one block range, (b) one kind of
rounding point, and prints the distance of the clip-normalised attention frames to the fp32 chain.  Usage:
    python tests/tools/vit_rounding_ablation.py [--frames 4] [--width 224] [--fmt f16]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import vit_ref_cpu as V   # noqa: E402


def block(sd, i, x, fmt, points, last):
    """block_forward of the oracle with the rounding points selectable: w (weights), ln (LayerNorm output), qkv, p (exponentiated
    probabilities), ao (attention output), hid (GELU output)."""
    def r(name):
        return V._rounder(fmt if name in points else None)
    p = f"blocks.{i}."
    b, n, _ = x.shape
    y = r("ln")(F.layer_norm(x, (V.DIM,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], V.LN_EPS))
    qkv = F.linear(y, r("w")(sd[p + "attn.qkv.weight"]), sd[p + "attn.qkv.bias"])
    qkv = r("qkv")(torch.cat([qkv[..., :V.DIM] * V.QSCALE, qkv[..., V.DIM:]], -1))
    qkv = qkv.reshape(b, n, 3, V.HEADS, V.DIM // V.HEADS).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    if last:
        s = q @ k.transpose(-2, -1)
        pexp = torch.exp2(s - s.max(-1, keepdim=True).values)
        return pexp / pexp.sum(-1, keepdim=True)
    y = r("ao")(V.flash_attention_emulated(q, k, v, r("p")).transpose(1, 2).reshape(b, n, V.DIM))
    x = x + F.linear(y, r("w")(sd[p + "attn.proj.weight"]), sd[p + "attn.proj.bias"])
    y = r("ln")(F.layer_norm(x, (V.DIM,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], V.LN_EPS))
    h = F.linear(y, r("w")(sd[p + "mlp.fc1.weight"]), sd[p + "mlp.fc1.bias"])
    y = r("hid")(V.gelu_poly(h))
    return x + F.linear(y, r("w")(sd[p + "mlp.fc2.weight"]), sd[p + "mlp.fc2.bias"])


ALL = ("w", "ln", "qkv", "p", "ao", "hid")


def maps(sd, frames, fmt, points, blocks):
    x = V.prepare_tokens(sd, frames, fmt if "patch" in points else None)
    for i in range(V.DEPTH):
        x = block(sd, i, x, fmt, points if i in blocks else (), i == V.DEPTH - 1)
    hw = frames.shape[-1] // V.PATCH
    return V.clip_normalise_ref(V.attention_frames_from_cls(x[:, :, 0, 1:], hw, hw))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--width", type=int, default=224)
    ap.add_argument("--fmt", default="f16")
    ap.add_argument("--seed", type=int, default=3)
    a = ap.parse_args()
    sd = V.seeded_vit_state(a.seed)
    frames = V.synthetic_frames(a.frames, a.width, 9)
    every = tuple(range(V.DEPTH))
    with torch.no_grad():
        ref = maps(sd, frames, None, (), ())
        cases = [("all points, all blocks", ALL + ("patch",), every)]
        cases += [(f"only '{k}', all blocks", (k,), every) for k in ALL + ("patch",)]
        cases += [("all points, blocks 0-10", ALL + ("patch",), every[:-1]), ("all points, block 11 only", ALL, (V.DEPTH - 1,)),
                  ("all points, blocks 0-5", ALL + ("patch",), every[:6]), ("all points, blocks 6-10", ALL, every[6:11])]
        print(f"# {a.frames} frames {a.width}^2, storage format {a.fmt}, seeded weights {a.seed}: clip-normalised attention frames vs fp32")
        for name, pts, blk in cases:
            e = (maps(sd, frames, a.fmt, pts, blk) - ref).abs()
            print(f"{name:28s} max|err| {e.max().item():.3e}  mean {e.mean().item():.3e}  rms {e.pow(2).mean().sqrt().item():.3e}", flush=True)


if __name__ == "__main__":
    main()
