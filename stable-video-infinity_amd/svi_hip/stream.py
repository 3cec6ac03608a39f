"""The clip loop of an image-conditioned stream (test_svi.py:424-476 around SVIVideoPipeline.__call__), resident on the GPU.

Per clip the reference: encodes the motion frames (the last `num_motion_frames` 8-bit frames of the previous clip, or the input
image) plus padding with the VAE into y (encode_images_adaptive, svi_video.py:291-364), denoises 50 steps, decodes, converts the
video to 8-bit PIL frames (tensor2video :366-370), keeps all but the last `num_motion_frames` of them (test_svi.py:472-476) and hands
those last frames — as 8-bit images, through preprocess_image — to the next clip.  Here the same sequence runs without leaving
HBM: the 8-bit quantisation and its inverse are device kernels with the reference's fp32 arithmetic (svi_video_to_u8 /
svi_u8_to_video), so the frames and the next clip's conditioning are what the host round trip would give.

Outside SURVEY §8's path and therefore injected: the prompt embeddings (T5) and the CLIP image feature of each clip's first frame
(`clip_encoder`, a callable [1, 3, H, W] float in [-1, 1] -> [1, 257, 1280]).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch

from . import _lib as L
from .conditioning import image_condition
from .pipeline import DenoiseLoop, generate_noise


def video_to_u8(video: torch.Tensor) -> torch.Tensor:
    """video fp32 [3, T, H, W] in [-1, 1] -> 8-bit frames [T, H, W, 3]   (tensor2video, svi_video.py:366-370)."""
    video = video.to(device="cuda", dtype=torch.float32).contiguous()
    _, T, H, W = video.shape
    out = torch.empty((T, H, W, 3), dtype=torch.uint8, device=video.device)
    L.check(L.lib().svi_video_to_u8(L.ptr(video), L.ptr(out), T, H, W, L.current_stream()), "svi_video_to_u8")
    return out


def u8_to_video(frames: torch.Tensor) -> torch.Tensor:
    """8-bit frames [n, H, W, 3] -> fp32 [n, 3, H, W] = x * (2 / 255) - 1   (preprocess_image, pipelines/base.py:44-45)."""
    frames = frames.to(device="cuda", dtype=torch.uint8).contiguous()
    n, H, W, _ = frames.shape
    out = torch.empty((n, 3, H, W), dtype=torch.float32, device=frames.device)
    L.check(L.lib().svi_u8_to_video(L.ptr(frames), L.ptr(out), n, H, W, L.current_stream()), "svi_u8_to_video")
    return out


class StreamLoop:
    def __init__(self, dit, vae, clip_encoder: Optional[Callable[[torch.Tensor], torch.Tensor]] = None, num_motion_frames: int = 1,
                 num_frames: int = 81, num_inference_steps: int = 50, cfg_scale: float = 5.0, sigma_shift: float = 5.0,
                 ref_pad_cfg: bool = False, ref_pad_num: int = 0, seed_times: int = 42, tiled: bool = False, tile_size=(30, 52),
                 tile_stride=(15, 26)):
        """tiled / tile_size / tile_stride: the VAE tiling arguments SVIVideoPipeline.__call__ passes to decode_video (svi_video.py:439-441, 515;
        test_svi.py passes args.tiled, default False).  The conditioning encode is never tiled, as in the reference (:350)."""
        if num_motion_frames < 1:
            raise ValueError("an image-conditioned stream hands at least one motion frame from clip to clip (test_svi.py:472-476)")
        self.loop = DenoiseLoop(dit, resident=True)      # every clip of the stream replays the step graph the first one captured
        self.vae, self.clip_encoder = vae, clip_encoder
        self.num_motion_frames, self.num_frames = num_motion_frames, num_frames
        self.steps, self.cfg_scale, self.sigma_shift = num_inference_steps, cfg_scale, sigma_shift
        self.ref_pad_cfg, self.ref_pad_num, self.seed_times = ref_pad_cfg, ref_pad_num, seed_times
        self.tiler = dict(tiled=tiled, tile_size=tuple(tile_size), tile_stride=tuple(tile_stride))
        self.trace: List[dict] = []                     # per clip: what conditioned it (for tests / inspection)

    @torch.no_grad()
    def run(self, input_frames_u8: torch.Tensor, ref_frame_u8: torch.Tensor, prompts: Sequence[tuple], num_clips: int,
            prompt_repeat_times: int = 1, use_first_prompt_only: bool = False, clip_feature: Optional[torch.Tensor] = None,
            start_clip: int = 0) -> torch.Tensor:
        """input_frames_u8 [n, H, W, 3] (the input image, n = 1, or the motion frames of an earlier stream), ref_frame_u8 [H, W, 3]
        (`random_ref_frame`), prompts: list of (context_pos, context_neg) embeddings.  Returns the stitched 8-bit video
        [frames, H, W, 3] on the GPU: every clip but the last loses its final `num_motion_frames` frames.
        `start_clip`: resume a stream at clip index k (input_frames_u8 = the motion frames clip k-1 handed over): clips
        k .. num_clips-1 run with the seeds and prompts of their own indices."""
        from .parallel import clip_prompt_index, clip_seed
        motion = input_frames_u8.to("cuda")
        ref = u8_to_video(ref_frame_u8.to("cuda")[None])[0]
        H, W = motion.shape[1:3]
        tlat = (self.num_frames - 1) // 4 + 1
        pieces = []
        self.trace = []
        cache_was_on = self.loop.dit._ctx_cache_on
        try:
            self._clips(range(start_clip, num_clips), num_clips, motion, ref, prompts, prompt_repeat_times, use_first_prompt_only, clip_feature, pieces, H, W, tlat)
        finally:
            if not cache_was_on:
                self.loop.close()      # the resident loop keeps the context cache on from clip to clip; hand the model back as it came
        return torch.cat(pieces, dim=0)

    def _clips(self, clips, num_clips, motion, ref, prompts, prompt_repeat_times, use_first_prompt_only, clip_feature, pieces, H, W, tlat) -> None:
        from .parallel import clip_prompt_index, clip_seed
        for k in clips:
            first = u8_to_video(motion)                                          # preprocess_image of every motion frame
            y = image_condition(self.vae, first, ref, self.num_frames, self.ref_pad_cfg, self.ref_pad_num)
            if self.clip_encoder is None:
                cf = clip_feature
            elif hasattr(self.clip_encoder, "encode_image"):                      # svi_hip.WanImageEncoder: clip_context of svi_video.py:317, :355
                cf = self.clip_encoder.encode_image([first[:1]]).to(torch.bfloat16)
            else:
                cf = self.clip_encoder(first[:1])
            ctx_pos, ctx_neg = prompts[clip_prompt_index(k, len(prompts), prompt_repeat_times, use_first_prompt_only)]
            seed = clip_seed(k, self.seed_times)
            lat = generate_noise((1, 16, tlat, H // 8, W // 8), seed=seed, device="cpu", dtype=torch.float32).to("cuda", torch.bfloat16)
            cond = dict(y=y)
            if cf is not None:
                cond["clip_feature"] = cf
            lat = self.loop.sample(lat, ctx_pos, ctx_neg, num_inference_steps=self.steps, cfg_scale=self.cfg_scale,
                                   sigma_shift=self.sigma_shift, **cond)
            video = self.vae.decode(lat.float(), device="cuda", **self.tiler)[0]  # [3, num_frames, H, W] fp32
            frames = video_to_u8(video)
            self.trace.append(dict(clip=k, seed=seed, motion=motion, y=y, latents=lat, frames=frames))
            motion = frames[-self.num_motion_frames:]
            pieces.append(frames[:-self.num_motion_frames] if k < num_clips - 1 else frames)
