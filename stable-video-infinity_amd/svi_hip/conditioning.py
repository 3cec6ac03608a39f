"""Image / motion-frame conditioning of an I2V clip, on the HIP VAE.

    image_condition(vae, first_frames, random_ref_frame, num_frames, ...)  <-  the tensor half of
        SVIVideoPipeline.encode_images_adaptive            pipelines/svi_video.py:291-364

The reference builds, per clip,  y = cat(mask[4, T', h, w], VAE.encode([motion frames ‖ padding])[16, T', h, w])  and hands
it to the DiT as 20 extra input channels (svi_video.py:346-350; models/wan_video_dit.py patch embedding in_dim = 36).
PIL resizing / normalisation (`preprocess_image`) stays with the caller: frames arrive as tensors in [-1, 1].  The CLIP tokens of the
first frame (`clip_context`, :317) come from svi_hip.WanImageEncoder when one is passed.  The reference flips the whole VAE to fp32 and back
around this call (:303-309, :359-362); the HIP VAE is fp32-resident, so nothing moves.
"""
from __future__ import annotations

from typing import Optional, Sequence, Union

import torch


def condition_mask(num_frames: int, height: int, width: int, num_condition_frames: int = 1, ref_pad_cfg: bool = False,
                   device="cuda", dtype=torch.float32) -> torch.Tensor:
    """Mask of svi_video.py:319-327 -> [4, 1 + (num_frames-1)//4, height//8, width//8]: ones on the conditioned frames
    (only the first frame unless ref_pad_cfg), the first frame repeated 4x, then folded 4 frames -> 4 channels."""
    msk = torch.ones(1, num_frames, height // 8, width // 8, device=device, dtype=dtype)
    if ref_pad_cfg:
        msk[:, num_condition_frames:] = 0
    else:
        msk[:, 1:] = 0
    msk = torch.concat([torch.repeat_interleave(msk[:, 0:1], repeats=4, dim=1), msk[:, 1:]], dim=1)
    msk = msk.view(1, msk.shape[1] // 4, 4, height // 8, width // 8)
    return msk.transpose(1, 2)[0]


def condition_video(first_frames: torch.Tensor, random_ref_frame: Optional[torch.Tensor], num_frames: int,
                    ref_pad_num: int = 0) -> torch.Tensor:
    """VAE input of svi_video.py:329-349: [3, num_frames, H, W] = motion frames followed by padding
    (ref_pad_num == 0: zeros; > 0: that many copies of the reference frame, then zeros; -1: the reference frame throughout).
    first_frames [n, 3, H, W], random_ref_frame [3, H, W] (or [1, 3, H, W])."""
    n, _, H, W = first_frames.shape
    remaining = num_frames - n
    cond = first_frames.permute(1, 0, 2, 3)
    if remaining <= 0:
        return cond[:, :num_frames].contiguous()
    if ref_pad_num == 0:
        pad = torch.zeros(3, remaining, H, W, device=first_frames.device, dtype=first_frames.dtype)
    else:
        if random_ref_frame is None:
            raise ValueError("ref_pad_num != 0 needs random_ref_frame")
        ref = random_ref_frame.reshape(3, 1, H, W).to(first_frames)
        if ref_pad_num == -1:
            pad = ref.repeat(1, remaining, 1, 1)
        else:
            k = min(ref_pad_num, remaining) if remaining > ref_pad_num else ref_pad_num
            parts = [ref] * k
            if remaining > ref_pad_num:
                parts += [torch.zeros(3, 1, H, W, device=first_frames.device, dtype=first_frames.dtype)] * (remaining - ref_pad_num)
            pad = torch.cat(parts, dim=1)
    return torch.concat([cond, pad], dim=1)


def image_condition(vae, first_frames: Union[torch.Tensor, Sequence[torch.Tensor]], random_ref_frame: Optional[torch.Tensor],
                    num_frames: int, ref_pad_cfg: bool = False, ref_pad_num: int = 0, out_dtype=torch.bfloat16,
                    tiled: bool = False, tile_size=(34, 34), tile_stride=(18, 16), image_encoder=None):
    """y [1, 20, T', H/8, W/8] exactly as encode_images_adaptive returns it (mask ‖ VAE latent, cast to the DiT dtype).
    With `image_encoder` (svi_hip.WanImageEncoder): the call's full result {"clip_feature": ..., "y": ...} (:317, :355-364) — the CLIP
    tokens of the FIRST motion frame, computed in fp32 and cast to the DiT dtype."""
    if not isinstance(first_frames, torch.Tensor):
        first_frames = torch.stack([f.reshape(3, *f.shape[-2:]) for f in first_frames])
    first_frames = first_frames.to(device="cuda", dtype=torch.float32)
    n, _, H, W = first_frames.shape
    video = condition_video(first_frames, random_ref_frame, num_frames, ref_pad_num)
    lat = vae.encode([video], device="cuda", tiled=tiled, tile_size=tile_size, tile_stride=tile_stride)[0]
    msk = condition_mask(num_frames, H, W, n, ref_pad_cfg, device=lat.device)
    y = torch.concat([msk, lat]).unsqueeze(0).to(out_dtype)
    if image_encoder is None:
        return y
    return {"clip_feature": image_encoder.encode_image([first_frames[:1]]).to(out_dtype), "y": y}
