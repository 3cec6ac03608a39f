"""svi_hip — MI355X-native backend for the Stable-Video-Infinity rolling-window denoising hot path.

Python is plumbing only (pointers, streams, torch.distributed); the arithmetic lives in libsvi_hip.so
(hand-written HIP for gfx950, C ABI in include/svi_hip.h).  Importing this package never imports the
oracle and never falls back to PyTorch math.
"""
from . import _lib
from .dit import WanDiT, model_fn_wan_talk_video, model_fn_wan_video
from .ops import cfg3_step_, cfg_step_, flash_attention, fp8_attention, fp8_attention_enabled, layernorm_modulate, linear, rmsnorm_rope_
from .pipeline import DenoiseLoop, generate_noise, install
from .scheduler import FlowMatchScheduler
from .vae import WanVideoVAE
from .conditioning import condition_mask, condition_video, image_condition
from .teacache import TeaCache
from .pose import PoseEmbedder
from .encoders import WanImageEncoder, WanTextEncoder
from . import checkpoint, lora, sequence_parallel
from .stream import StreamLoop, u8_to_video, video_to_u8

__all__ = ["WanDiT", "model_fn_wan_video", "model_fn_wan_talk_video", "cfg3_step_", "flash_attention", "fp8_attention", "fp8_attention_enabled", "layernorm_modulate", "rmsnorm_rope_", "linear",
           "cfg_step_", "DenoiseLoop", "generate_noise", "install", "FlowMatchScheduler", "WanVideoVAE", "condition_mask", "condition_video", "image_condition", "TeaCache", "PoseEmbedder", "WanTextEncoder", "WanImageEncoder", "StreamLoop", "video_to_u8", "u8_to_video", "_lib"]
