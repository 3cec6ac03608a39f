"""Time the prompt-side encoders at the real model sizes on the GPU (random weights): umT5-XXL (24 layers, 512 positions) and the
ViT-H/14 visual tower (31 of 32 blocks, one 480x832 frame).  They run once per clip (twice for the text encoder: positive and
negative prompt), outside the step loop.
    python tools/encoder_timing.py
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-video-infinity_amd"))
import svi_hip                                                                                       # noqa: E402
from svi_hip import _lib as L                                                                          # noqa: E402


def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    dim, da, df, heads, layers = 4096, 4096, 10240, 64, 24
    sd = {"token_embedding.weight": torch.randn(256384, dim, device=dev, generator=g, dtype=torch.bfloat16), "norm.weight": torch.ones(dim, device=dev, dtype=torch.bfloat16)}
    for i in range(layers):
        p = f"blocks.{i}."
        for n, shp in (("attn.q", (da, dim)), ("attn.k", (da, dim)), ("attn.v", (da, dim)), ("attn.o", (dim, da)), ("ffn.gate.0", (df, dim)),
                       ("ffn.fc1", (df, dim)), ("ffn.fc2", (dim, df))):
            sd[p + n + ".weight"] = (torch.randn(shp, device=dev, generator=g, dtype=torch.bfloat16) * shp[1] ** -0.5)
        sd[p + "norm1.weight"] = torch.ones(dim, device=dev, dtype=torch.bfloat16)
        sd[p + "norm2.weight"] = torch.ones(dim, device=dev, dtype=torch.bfloat16)
        sd[p + "pos_embedding.embedding.weight"] = torch.randn(32, heads, device=dev, generator=g, dtype=torch.bfloat16)
    te = svi_hip.WanTextEncoder.from_state_dict(sd)
    params = sum(v.numel() for k, v in sd.items() if k != "token_embedding.weight")
    for valid in (40, 128, 512):
        ids = torch.randint(1, 256384, (1, 512), device=dev)
        ids[0, valid:] = 0
        mask = (torch.arange(512)[None] < valid).long()
        ms_v = timed(lambda: te.forward(ids, mask, rows="valid"))
        ms_a = timed(lambda: te.forward(ids, mask, rows="all"))
        print(f"umT5-XXL encoder ({params / 1e9:.2f} B block parameters, bf16): {valid:3d} valid tokens of 512 -> {ms_v:7.2f} ms valid rows only "
              f"({2 * params * valid / ms_v / 1e9:6.1f} TFLOP/s, weights {2 * params / ms_v / 1e6:5.0f} GB/s), {ms_a:7.2f} ms all 512 rows")
        assert bool(torch.isfinite(te.forward(ids, mask).float()).all())
    del te, sd
    torch.cuda.empty_cache()
    dim, heads, layers = 1280, 16, 32
    sd = {"cls_embedding": torch.randn(1, 1, dim, device=dev, generator=g) * dim ** -0.5, "pos_embedding": torch.randn(1, 257, dim, device=dev, generator=g) * dim ** -0.5,
          "patch_embedding.weight": torch.randn(dim, 3, 14, 14, device=dev, generator=g) * 588 ** -0.5,
          "pre_norm.weight": torch.ones(dim, device=dev), "pre_norm.bias": torch.zeros(dim, device=dev)}
    for i in range(layers):
        p = f"transformer.{i}."
        for n, shp in (("attn.to_qkv", (3 * dim, dim)), ("attn.proj", (dim, dim)), ("mlp.0", (4 * dim, dim)), ("mlp.2", (dim, 4 * dim))):
            sd[p + n + ".weight"] = torch.randn(shp, device=dev, generator=g) * shp[1] ** -0.5
            sd[p + n + ".bias"] = torch.zeros(shp[0], device=dev)
        for n in ("norm1", "norm2"):
            sd[p + n + ".weight"] = torch.ones(dim, device=dev)
            sd[p + n + ".bias"] = torch.zeros(dim, device=dev)
    ie = svi_hip.WanImageEncoder.from_state_dict(sd, num_heads=heads)
    img = torch.rand(1, 3, 480, 832, device=dev) * 2 - 1
    ms = timed(lambda: ie.encode_image([img]))
    flop = 2 * 257 * 31 * (12 * dim * dim) + 4 * 31 * 257 * 257 * dim
    print(f"CLIP ViT-H/14 visual tower, 31 blocks, fp32 (exact-fp32 MFMA): one 480x832 frame -> {ms:.2f} ms ({flop / ms / 1e9:.1f} TFLOP/s fp32)")
    assert bool(torch.isfinite(ie.encode_image([img])).all())


if __name__ == "__main__":
    main()
