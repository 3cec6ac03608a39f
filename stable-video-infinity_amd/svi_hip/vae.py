"""WanVideoVAE on libsvi_hip: same encode/decode signatures as the reference class
(diffsynth/models/wan_video_vae.py:599-789).

    WanVideoVAE.from_state_dict(sd)    <- WanVideoVAE().load_state_dict(sd)
    WanVideoVAE.from_module(ref_vae)   <- borrow the fp32 parameters of a reference WanVideoVAE on the GPU
    .encode(videos, device, tiled=False, tile_size, tile_stride) -> Tensor[N,16,T',h,w]      vae:759-774
    .decode(hidden_states, device, tiled=False, tile_size, tile_stride) -> Tensor[N,3,T,H,W] vae:777-789

The whole clip stays resident in HBM (no temporal chunking, no feature cache: see csrc/svi_vae.hip).  `tiled=True`
is the reference's spatial tiling and linear-ramp blending (vae:621-744) in one C call (svi_vae_tiled_decode / _encode): tiles are read
in place, decoded by the same kernels and blended on the GPU with the reference's fp32 arithmetic and task order.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Sequence

import torch

from . import _lib as L


def vae_param_shapes() -> Dict[str, tuple]:
    """Parameter inventory of the reference architecture (dim 96, z 16, mult 1,2,4,4; vae:494-517)."""
    s: Dict[str, tuple] = {}

    def conv3(n, o, i, k=(3, 3, 3)):
        s[n + ".weight"] = (o, i, *k); s[n + ".bias"] = (o,)

    def conv2(n, o, i, k=3):
        s[n + ".weight"] = (o, i, k, k); s[n + ".bias"] = (o,)

    def res(p, i, o):
        s[p + "residual.0.gamma"] = (i, 1, 1, 1); conv3(p + "residual.2", o, i)
        s[p + "residual.3.gamma"] = (o, 1, 1, 1); conv3(p + "residual.6", o, o)
        if i != o:
            conv3(p + "shortcut", o, i, (1, 1, 1))

    def attn(p, c):
        s[p + "norm.gamma"] = (c, 1, 1); conv2(p + "to_qkv", 3 * c, c, 1); conv2(p + "proj", c, c, 1)

    base, z, mult, tdown = 96, 16, (1, 2, 4, 4), (False, True, True)
    e, d = "model.encoder.", "model.decoder."
    dims = [base * u for u in (1,) + mult]
    conv3(e + "conv1", dims[0], 3)
    idx = 0
    for i, (di, do) in enumerate(zip(dims[:-1], dims[1:])):
        for _ in range(2):
            res(f"{e}downsamples.{idx}.", di, do); idx += 1; di = do
        if i != 3:
            conv2(f"{e}downsamples.{idx}.resample.1", do, do)
            if tdown[i]:
                conv3(f"{e}downsamples.{idx}.time_conv", do, do, (3, 1, 1))
            idx += 1
    top = dims[-1]
    res(e + "middle.0.", top, top); attn(e + "middle.1.", top); res(e + "middle.2.", top, top)
    s[e + "head.0.gamma"] = (top, 1, 1, 1); conv3(e + "head.2", 2 * z, top)
    conv3("model.conv1", 2 * z, 2 * z, (1, 1, 1)); conv3("model.conv2", z, z, (1, 1, 1))
    dims = [base * u for u in (mult[-1],) + mult[::-1]]
    conv3(d + "conv1", dims[0], z)
    res(d + "middle.0.", dims[0], dims[0]); attn(d + "middle.1.", dims[0]); res(d + "middle.2.", dims[0], dims[0])
    idx, tup = 0, tdown[::-1]
    for i, (di, do) in enumerate(zip(dims[:-1], dims[1:])):
        if i >= 1:
            di //= 2
        for _ in range(3):
            res(f"{d}upsamples.{idx}.", di, do); idx += 1; di = do
        if i != 3:
            conv2(f"{d}upsamples.{idx}.resample.1", do // 2, do)
            if tup[i]:
                conv3(f"{d}upsamples.{idx}.time_conv", do * 2, do, (3, 1, 1))
            idx += 1
    s[d + "head.0.gamma"] = (dims[-1], 1, 1, 1); conv3(d + "head.2", 3, dims[-1])
    return s


def device_vae_weights(seed: int, device) -> Dict[str, torch.Tensor]:
    """Random-init fp32 weights of the VAE architecture, generated on the GPU (bench / smoke only)."""
    g = torch.Generator(device=device).manual_seed(seed)
    out = {}
    for name, shape in vae_param_shapes().items():
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "gamma":
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device)
        elif leaf == "weight":
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = (torch.rand(shape, generator=g, device=device) * 2 - 1) / math.sqrt(fan_in)
        else:
            t = (torch.rand(shape, generator=g, device=device) * 2 - 1) * 0.05
        out[name] = t.float().contiguous()
    return out


class WanVideoVAE:
    upsampling_factor = 8

    def __init__(self):
        h = C.c_void_p()
        L.check(L.lib().svi_vae_create(C.byref(h)), "svi_vae_create")
        self._h = h
        self._params: Dict[str, torch.Tensor] = {}

    @classmethod
    def from_state_dict(cls, state_dict: Dict[str, torch.Tensor], device="cuda") -> "WanVideoVAE":
        v = cls()
        v.bind({k: t.to(device=device, dtype=torch.float32).contiguous() for k, t in state_dict.items()})
        return v

    @classmethod
    def from_module(cls, ref_vae) -> "WanVideoVAE":
        v = cls()
        v.bind({k: t.to(device="cuda", dtype=torch.float32).contiguous() for k, t in ref_vae.state_dict().items()})
        return v

    def bind(self, state_dict: Dict[str, torch.Tensor]) -> None:
        lib = L.lib()
        for name, t in state_dict.items():
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError(f"VAE parameter {name} must be a contiguous CUDA fp32 tensor")
            shape = (C.c_int64 * t.dim())(*t.shape)
            L.check(lib.svi_vae_bind_weight(self._h, name.encode(), t.data_ptr(), L.SVI_F32, shape, t.dim()), f"bind {name}")
            self._params[name] = t
        L.check(lib.svi_vae_check_bound(self._h), "svi_vae_check_bound")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.lib().svi_vae_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- single (whole-frame) paths: vae:747-756 ----------------------------------------------------------------
    def single_decode(self, hidden_state: torch.Tensor, device=None) -> torch.Tensor:
        """[1,16,T,h,w] -> [1,3,1+4(T-1),8h,8w] fp32, clamped to [-1,1]."""
        z = hidden_state.to(device="cuda", dtype=torch.float32).contiguous()
        _, c, t, h, w = z.shape
        if c != 16:
            raise ValueError("latents must have 16 channels")
        out = torch.empty((1, 3, 1 + 4 * (t - 1), 8 * h, 8 * w), dtype=torch.float32, device=z.device)
        L.check(L.lib().svi_vae_decode(self._h, L.ptr(z), L.ptr(out), t, h, w, L.current_stream()), "svi_vae_decode")
        return out

    def single_encode(self, video: torch.Tensor, device=None) -> torch.Tensor:
        """[1,3,1+4k,H,W] -> [1,16,1+k,H/8,W/8] fp32."""
        v = video.to(device="cuda", dtype=torch.float32).contiguous()
        _, c, t, h, w = v.shape
        if c != 3:
            raise ValueError("video must have 3 channels")
        out = torch.empty((1, 16, 1 + (t - 1) // 4, h // 8, w // 8), dtype=torch.float32, device=v.device)
        L.check(L.lib().svi_vae_encode(self._h, L.ptr(v), L.ptr(out), t, h, w, L.current_stream()), "svi_vae_encode")
        return out

    # ---- spatial tiling: vae:621-744 (one C call each; tiles, masks and the blend run on the device) ---------------------
    def tiled_decode(self, hidden_states: torch.Tensor, device, tile_size, tile_stride) -> torch.Tensor:
        """[1,16,T,h,w] -> [1,3,4T-3,8h,8w]; tile_size / tile_stride in latent pixels (vae:643-693)."""
        z = hidden_states.to(device="cuda", dtype=torch.float32).contiguous()
        _, c, t, h, w = z.shape
        if c != 16:
            raise ValueError("latents must have 16 channels")
        out = torch.empty((1, 3, 4 * t - 3, 8 * h, 8 * w), dtype=torch.float32, device=z.device)
        L.check(L.lib().svi_vae_tiled_decode(self._h, L.ptr(z), L.ptr(out), t, h, w, int(tile_size[0]), int(tile_size[1]),
                                             int(tile_stride[0]), int(tile_stride[1]), L.current_stream()), "svi_vae_tiled_decode")
        return out

    def tiled_encode(self, video: torch.Tensor, device, tile_size, tile_stride) -> torch.Tensor:
        """[1,3,T,H,W] -> [1,16,(T+3)//4,H/8,W/8]; tile_size / tile_stride in video pixels (vae:696-744)."""
        v = video.to(device="cuda", dtype=torch.float32).contiguous()
        _, c, t, h, w = v.shape
        if c != 3:
            raise ValueError("video must have 3 channels")
        out = torch.empty((1, 16, (t + 3) // 4, h // 8, w // 8), dtype=torch.float32, device=v.device)
        L.check(L.lib().svi_vae_tiled_encode(self._h, L.ptr(v), L.ptr(out), t, h, w, int(tile_size[0]), int(tile_size[1]),
                                             int(tile_stride[0]), int(tile_stride[1]), L.current_stream()), "svi_vae_tiled_encode")
        return out

    # ---- public surface: vae:759-789 ----------------------------------------------------------------------------------
    def encode(self, videos: Sequence[torch.Tensor], device=None, tiled=False, tile_size=(34, 34), tile_stride=(18, 16)):
        outs: List[torch.Tensor] = []
        for video in videos:
            video = video.unsqueeze(0)
            if tiled:
                # as the reference (vae:765-767): the x8 is applied to the loop's own variables, so from the second video of a
                # batch on the tiles are 8x larger again (one tile per video in practice); kept, it is observable behaviour
                tile_size = (tile_size[0] * 8, tile_size[1] * 8)
                tile_stride = (tile_stride[0] * 8, tile_stride[1] * 8)
                hs = self.tiled_encode(video, device, tile_size, tile_stride)
            else:
                hs = self.single_encode(video, device)
            outs.append(hs.squeeze(0))
        return torch.stack(outs)

    def decode(self, hidden_states, device=None, tiled=False, tile_size=(34, 34), tile_stride=(18, 16)):
        outs: List[torch.Tensor] = []
        for hs in hidden_states:
            hs = hs.unsqueeze(0)
            if tiled:
                video = self.tiled_decode(hs, device, tile_size, tile_stride)
            else:
                video = self.single_decode(hs, device)
            outs.append(video.squeeze(0))
        return torch.stack(outs)
