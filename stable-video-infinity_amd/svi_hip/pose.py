"""Dance variant: the pose embedder of SVIDanceVideoPipeline on libsvi_hip.

    PoseEmbedder.from_state_dict(sd)        <- pipe.dwpose_embedding.load_state_dict(sd)     pipelines/svi_video_dance.py:255-275
    PoseEmbedder.from_module(seq)           <- an existing reference nn.Sequential (fp32 parameters borrowed)
    embedder(humanpose_data)                <- :527-530: cat(first frame x3, pose) / 255 -> dwpose_embedding -> bf16 ->
                                               rearrange 'b c f h w -> b (f h w) c'   = the `add_condition` of model_fn_wan_video

The seven Conv3d + SiLU layers run on the exact-fp32 MFMA convolution of the VAE (csrc/svi_vae.hip); the last one stores the bf16
token rows directly.  In the sampler the condition goes to the CONDITIONAL forward only (the unconditional one gets None unless
`cond_wo_pose`, :423-430) — DenoiseLoop.sample(add_condition=..., cond_wo_pose=...) does the same.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import torch

from . import _lib as L


class PoseEmbedder:
    def __init__(self, hidden: int = 16, dim: int = 5120):
        self.hidden, self.dim = hidden, dim
        h = C.c_void_p()
        L.check(L.lib().svi_pose_create(hidden, dim, C.byref(h)), "svi_pose_create")
        self._h = h
        self._params: Dict[str, torch.Tensor] = {}

    @classmethod
    def from_state_dict(cls, state_dict: Dict[str, torch.Tensor], device="cuda") -> "PoseEmbedder":
        sd = {k.split("dwpose_embedding.")[-1]: v for k, v in state_dict.items()}          # the checkpoint's prefix (dance:272-273)
        m = cls(hidden=sd["0.weight"].shape[0], dim=sd["12.weight"].shape[0])
        m.bind({k: v.to(device=device, dtype=torch.float32).contiguous() for k, v in sd.items()})
        return m

    @classmethod
    def from_module(cls, seq) -> "PoseEmbedder":
        return cls.from_state_dict(dict(seq.state_dict()))

    def bind(self, state_dict: Dict[str, torch.Tensor]) -> None:
        lib = L.lib()
        for name, t in state_dict.items():
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError(f"pose embedder parameter {name} must be a contiguous CUDA fp32 tensor")
            shape = (C.c_int64 * t.dim())(*t.shape)
            L.check(lib.svi_pose_bind_weight(self._h, name.encode(), t.data_ptr(), L.SVI_F32, shape, t.dim()), f"bind {name}")
            self._params[name] = t
        L.check(lib.svi_pose_check_bound(self._h), "svi_pose_check_bound")

    def tokens(self, F: int, H: int, W: int) -> Tuple[int, int, int]:
        f, h, w = C.c_int32(), C.c_int32(), C.c_int32()
        L.check(L.lib().svi_pose_tokens(self._h, F, H, W, C.byref(f), C.byref(h), C.byref(w)), "svi_pose_tokens")
        return f.value, h.value, w.value

    def __call__(self, humanpose_data: torch.Tensor) -> torch.Tensor:
        """humanpose_data [3, F, H, W] (0..255, any float / integer dtype) -> add_condition bf16 [1, f*h*w, dim]."""
        if humanpose_data.dim() != 4 or humanpose_data.shape[0] != 3:
            raise ValueError(f"humanpose_data must be [3, F, H, W] (got {tuple(humanpose_data.shape)})")
        x = humanpose_data.to(device="cuda", dtype=torch.float32).contiguous()
        _, F, H, W = x.shape
        f, h, w = self.tokens(F, H, W)
        out = torch.empty((1, f * h * w, self.dim), dtype=torch.bfloat16, device=x.device)
        L.check(L.lib().svi_pose_forward(self._h, L.ptr(x), L.ptr(out), F, H, W, L.current_stream()), "svi_pose_forward")
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.lib().svi_pose_destroy(self._h)
                self._h = None
        except Exception:
            pass
