"""CPU oracle for the Wan 3-D causal VAE encode/decode — TEST INFRASTRUCTURE ONLY
(see oracle/wan_dit_oracle.py header for the import rule).

Parity status: PINNED.  tests/golden/vae_*.npz hold outputs of the reference's own
`WanVideoVAE.encode/decode` (chunked, cached implementation; imported by tests/gen_golden.py)
on seeded weights/inputs; tests/test_oracle_vae.py checks this file against them.

The reference streams the video through the network in temporal chunks (encode: frame 0, then
4-frame chunks; decode: one latent frame at a time) and gives every causal conv a 2-frame
feature cache (diffsynth/models/wan_video_vae.py:33-52, 198-232, 525-575).  Unrolled over the
whole clip that is exactly:
  * every CausalConv3d = a conv over the full sequence with (kt-1) zero frames in front;
  * decoder `upsample3d` (:122-156): frame 0 skips the time conv (the 'Rep' sentinel) and is
    hidden from later frames' history; frame t>=1 -> two frames, the two channel halves of
    time_conv's output;
  * encoder `downsample3d` (:162-173): frame 0 passes through; then a stride-2, k=3,
    un-padded time conv over the full sequence (windows start at frame 0).
This oracle states that whole-sequence form; the HIP path streams like the reference.  Both
must agree with the golden vectors.

Restates: wan_video_vae.py:55-70 RMS_norm, :82-174 Resample, :198-232 ResidualBlock,
:235-273 AttentionBlock, :276-376 Encoder3d, :379-481 Decoder3d, :525-575 encode/decode,
:604-614 latent mean/std, :753-756 clamp.
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

LATENT_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
               0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
LATENT_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
              3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]

BASE_DIM = 96
Z_DIM = 16
DIM_MULT = (1, 2, 4, 4)
NUM_RES = 2
TEMPORAL_DOWN = (False, True, True)


# -------------------------------------------------------------------------------------------
def causal_conv3d(x: Tensor, w: Tensor, b: Tensor, stride_t: int = 1, front_pad: bool = True) -> Tensor:
    """x [B,C,T,H,W]; spatial 'same' padding, temporal padding only in front."""
    kt, kh, kw = w.shape[2:]
    pt = (kt - 1) if front_pad else 0
    x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2, pt, 0))
    return F.conv3d(x, w, b, stride=(stride_t, 1, 1))


def channel_rms(x: Tensor, gamma: Tensor) -> Tensor:
    """x / max(||x||_2 over channels, 1e-12) * sqrt(C) * gamma (vae:55-70); x [B,C,...]."""
    c = x.shape[1]
    nrm = x.pow(2).sum(dim=1, keepdim=True).sqrt().clamp_min(1e-12)
    return x / nrm * math.sqrt(c) * gamma.reshape(1, c, *([1] * (x.dim() - 2)))


def silu(x: Tensor) -> Tensor:
    return x * torch.sigmoid(x)


def residual_block(sd: Dict[str, Tensor], p: str, x: Tensor) -> Tensor:
    skip = x
    if (p + "shortcut.weight") in sd:
        skip = causal_conv3d(x, sd[p + "shortcut.weight"], sd[p + "shortcut.bias"])
    h = silu(channel_rms(x, sd[p + "residual.0.gamma"]))
    h = causal_conv3d(h, sd[p + "residual.2.weight"], sd[p + "residual.2.bias"])
    h = silu(channel_rms(h, sd[p + "residual.3.gamma"]))
    h = causal_conv3d(h, sd[p + "residual.6.weight"], sd[p + "residual.6.bias"])
    return h + skip


def attention_block(sd: Dict[str, Tensor], p: str, x: Tensor) -> Tensor:
    """Per-frame single-head spatial self-attention, head dim = C (vae:235-273)."""
    b, c, t, h, w = x.shape
    fr = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    n = channel_rms(fr, sd[p + "norm.gamma"])
    qkv = F.conv2d(n, sd[p + "to_qkv.weight"], sd[p + "to_qkv.bias"]).reshape(b * t, 3 * c, h * w).transpose(1, 2)
    q, k, v = qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]
    att = torch.softmax((q @ k.transpose(1, 2)) / math.sqrt(c), dim=-1) @ v          # [bt, hw, c]
    o = F.conv2d(att.transpose(1, 2).reshape(b * t, c, h, w), sd[p + "proj.weight"], sd[p + "proj.bias"])
    return o.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4) + x


def per_frame_conv2d(x: Tensor, w: Tensor, b: Tensor, stride: int, pad) -> Tensor:
    bsz, c, t, h, wd = x.shape
    fr = x.permute(0, 2, 1, 3, 4).reshape(bsz * t, c, h, wd)
    fr = F.pad(fr, pad)
    out = F.conv2d(fr, w, b, stride=stride)
    return out.reshape(bsz, t, out.shape[1], out.shape[2], out.shape[3]).permute(0, 2, 1, 3, 4)


def upsample_block(sd: Dict[str, Tensor], p: str, x: Tensor, temporal: bool) -> Tensor:
    if temporal and x.shape[2] > 1:
        c = x.shape[1]
        hist = x.clone()
        hist[:, :, 0] = 0                                   # frame 0 is invisible to the time conv
        tc = causal_conv3d(hist, sd[p + "time_conv.weight"], sd[p + "time_conv.bias"])[:, :, 1:]
        pair = torch.stack([tc[:, :c], tc[:, c:]], dim=3)   # [B,C,T-1,2,H,W]
        x = torch.cat([x[:, :, :1], pair.reshape(x.shape[0], c, -1, x.shape[3], x.shape[4])], dim=2)
    b, c, t, h, w = x.shape
    fr = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    fr = fr.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)      # nearest-exact x2
    out = F.conv2d(fr, sd[p + "resample.1.weight"], sd[p + "resample.1.bias"], padding=1)
    return out.reshape(b, t, out.shape[1], 2 * h, 2 * w).permute(0, 2, 1, 3, 4)


def downsample_block(sd: Dict[str, Tensor], p: str, x: Tensor, temporal: bool) -> Tensor:
    x = per_frame_conv2d(x, sd[p + "resample.1.weight"], sd[p + "resample.1.bias"], 2, (0, 1, 0, 1))
    if temporal and x.shape[2] > 1:
        tc = causal_conv3d(x, sd[p + "time_conv.weight"], sd[p + "time_conv.bias"], stride_t=2, front_pad=False)
        x = torch.cat([x[:, :, :1], tc], dim=2)
    return x


# -------------------------------------------------------------------------------------------
def decoder_layout() -> List[tuple]:
    """(kind, index) list of decoder.upsamples entries for dim=96, mult (1,2,4,4) (vae:409-426)."""
    out, idx = [], 0
    temporal_up = TEMPORAL_DOWN[::-1]
    for i in range(len(DIM_MULT)):
        for _ in range(NUM_RES + 1):
            out.append(("res", idx)); idx += 1
        if i != len(DIM_MULT) - 1:
            out.append(("up3d" if temporal_up[i] else "up2d", idx)); idx += 1
    return out


def encoder_layout() -> List[tuple]:
    out, idx = [], 0
    for i in range(len(DIM_MULT)):
        for _ in range(NUM_RES):
            out.append(("res", idx)); idx += 1
        if i != len(DIM_MULT) - 1:
            out.append(("down3d" if TEMPORAL_DOWN[i] else "down2d", idx)); idx += 1
    return out


def vae_decode(sd: Dict[str, Tensor], z: Tensor, clamp: bool = True) -> Tensor:
    """latents [B,16,T,h,w] (normalised) -> video [B,3,1+4(T-1),8h,8w] clamped to [-1,1] (single_decode, vae:753-756;
    clamp=False: VideoVAE_.decode alone, which is what the tiled path blends, vae:666)."""
    mean = torch.tensor(LATENT_MEAN, dtype=z.dtype).reshape(1, Z_DIM, 1, 1, 1)
    inv_std = (1.0 / torch.tensor(LATENT_STD)).to(z.dtype).reshape(1, Z_DIM, 1, 1, 1)
    x = z / inv_std + mean
    x = causal_conv3d(x, sd["model.conv2.weight"], sd["model.conv2.bias"])
    d = "model.decoder."
    x = causal_conv3d(x, sd[d + "conv1.weight"], sd[d + "conv1.bias"])
    x = residual_block(sd, d + "middle.0.", x)
    x = attention_block(sd, d + "middle.1.", x)
    x = residual_block(sd, d + "middle.2.", x)
    for kind, i in decoder_layout():
        p = f"{d}upsamples.{i}."
        x = residual_block(sd, p, x) if kind == "res" else upsample_block(sd, p, x, kind == "up3d")
    x = silu(channel_rms(x, sd[d + "head.0.gamma"]))
    x = causal_conv3d(x, sd[d + "head.2.weight"], sd[d + "head.2.bias"])
    return x.clamp(-1.0, 1.0) if clamp else x


def vae_encode(sd: Dict[str, Tensor], video: Tensor) -> Tensor:
    """video [B,3,1+4k,H,W] in [-1,1] -> normalised latent mean [B,16,1+k,H/8,W/8]."""
    e = "model.encoder."
    x = causal_conv3d(video, sd[e + "conv1.weight"], sd[e + "conv1.bias"])
    for kind, i in encoder_layout():
        p = f"{e}downsamples.{i}."
        x = residual_block(sd, p, x) if kind == "res" else downsample_block(sd, p, x, kind == "down3d")
    x = residual_block(sd, e + "middle.0.", x)
    x = attention_block(sd, e + "middle.1.", x)
    x = residual_block(sd, e + "middle.2.", x)
    x = silu(channel_rms(x, sd[e + "head.0.gamma"]))
    x = causal_conv3d(x, sd[e + "head.2.weight"], sd[e + "head.2.bias"])
    x = causal_conv3d(x, sd["model.conv1.weight"], sd["model.conv1.bias"])
    mu = x[:, :Z_DIM]
    mean = torch.tensor(LATENT_MEAN, dtype=mu.dtype).reshape(1, Z_DIM, 1, 1, 1)
    inv_std = (1.0 / torch.tensor(LATENT_STD)).to(mu.dtype).reshape(1, Z_DIM, 1, 1, 1)
    return (mu - mean) * inv_std


# ---- spatial tiling (WanVideoVAE.tiled_decode / tiled_encode, vae:621-744) ---------------------------------------------------
def ramp_1d(length: int, left_bound: bool, right_bound: bool, border: int) -> Tensor:
    """build_1d_mask (vae:621-627)."""
    x = torch.ones(length)
    if not left_bound:
        x[:border] = (torch.arange(border) + 1) / border
    if not right_bound:
        x[-border:] = torch.flip((torch.arange(border) + 1) / border, dims=(0,))
    return x


def tile_tasks(H: int, W: int, size, stride) -> List[tuple]:
    """The task list of vae:648-655: a start is dropped when the previous tile already reaches the far edge."""
    out = []
    for h in range(0, H, stride[0]):
        if h - stride[0] >= 0 and h - stride[0] + size[0] >= H:
            continue
        for w in range(0, W, stride[1]):
            if w - stride[1] >= 0 and w - stride[1] + size[1] >= W:
                continue
            out.append((h, h + size[0], w, w + size[1]))
    return out


def _blend(tiles_fn, src: Tensor, out_shape, size, stride, scale_to_out, border) -> Tensor:
    _, _, _, H, W = src.shape
    values = torch.zeros(out_shape, dtype=src.dtype)
    weight = torch.zeros((1, 1, *out_shape[2:]), dtype=src.dtype)
    for h, h_, w, w_ in tile_tasks(H, W, size, stride):
        tile = tiles_fn(src[:, :, :, h:h_, w:w_])
        th, tw = tile.shape[3:]
        m = torch.minimum(ramp_1d(th, h == 0, h_ >= H, border[0])[:, None].expand(th, tw),
                          ramp_1d(tw, w == 0, w_ >= W, border[1])[None, :].expand(th, tw))[None, None, None]      # build_mask :630-640
        oh, ow = scale_to_out(h), scale_to_out(w)
        values[:, :, :, oh:oh + th, ow:ow + tw] += tile * m                                                   # :668-676
        weight[:, :, :, oh:oh + th, ow:ow + tw] += m                                                          # :677-685
    return values / weight


def tiled_decode(sd: Dict[str, Tensor], z: Tensor, tile_size, tile_stride) -> Tensor:
    """vae:643-693; sizes in latent pixels; un-clamped tiles, one clamp after the blend (:687)."""
    B, _, T, H, W = z.shape
    out = _blend(lambda t: vae_decode(sd, t, clamp=False), z, (B, 3, 4 * T - 3, 8 * H, 8 * W), tile_size, tile_stride, lambda a: a * 8,
                 ((tile_size[0] - tile_stride[0]) * 8, (tile_size[1] - tile_stride[1]) * 8))
    return out.clamp(-1.0, 1.0)


def tiled_encode(sd: Dict[str, Tensor], video: Tensor, tile_size, tile_stride) -> Tensor:
    """vae:696-744; sizes in VIDEO pixels (WanVideoVAE.encode multiplies its latent-unit arguments by 8 first, :765-767)."""
    B, _, T, H, W = video.shape
    return _blend(lambda t: vae_encode(sd, t), video, (B, 16, (T + 3) // 4, H // 8, W // 8), tile_size, tile_stride, lambda a: a // 8,
                  ((tile_size[0] - tile_stride[0]) // 8, (tile_size[1] - tile_stride[1]) // 8))


# ---- I2V conditioning assembly (pipelines/svi_video.py:313-350), the tensor half of encode_images_adaptive ------------------
def image_condition(sd: Dict[str, Tensor], first_frames: Tensor, random_ref_frame, num_frames: int, ref_pad_cfg: bool = False,
                    ref_pad_num: int = 0) -> Tensor:
    """first_frames [n,3,H,W] in [-1,1] -> y [1,20,T',H/8,W/8] fp32 = cat(mask, vae_encode([motion frames | padding])).
    Pinned: tests/golden/image_condition.npz holds y as returned by the reference's own encode_images_adaptive (compiled out
    of pipelines/svi_video.py by tests/gen_golden.py, run on a stand-in pipeline object with the seeded reference VAE);
    tests/test_conditioning.py checks this restatement against it.  Line map: mask :319-327, conditioned frames :329-335,
    padding :337-347, encode + concat :349-351."""
    n, _, H, W = first_frames.shape
    msk = torch.ones(1, num_frames, H // 8, W // 8)
    if ref_pad_cfg:
        msk[:, n:] = 0                                                               # :321
    else:
        msk[:, 1:] = 0                                                               # :323
    msk = torch.concat([torch.repeat_interleave(msk[:, 0:1], repeats=4, dim=1), msk[:, 1:]], dim=1)   # :324
    msk = msk.view(1, msk.shape[1] // 4, 4, H // 8, W // 8).transpose(1, 2)[0]       # :325-326
    cond = first_frames.permute(1, 0, 2, 3)                                           # :333 / :335
    remaining = num_frames - n
    if ref_pad_num == 0:
        pad = torch.zeros(3, remaining, H, W)                                         # :338
    elif ref_pad_num == -1:
        pad = random_ref_frame.reshape(3, 1, H, W).repeat(1, remaining, 1, 1)         # :347
    else:
        parts = [random_ref_frame.reshape(3, 1, H, W)] * ref_pad_num                  # :342-343
        if remaining > ref_pad_num:
            parts += [torch.zeros(3, 1, H, W)] * (remaining - ref_pad_num)            # :344-345
        pad = torch.cat(parts, dim=1)
    video = torch.concat([cond, pad], dim=1)                                          # :348
    y = vae_encode(sd, video.unsqueeze(0))[0]                                         # :349
    return torch.concat([msk, y]).unsqueeze(0)                                        # :350-351
