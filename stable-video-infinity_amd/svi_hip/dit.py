"""Host-side mirror of the reference DiT call surface, running on libsvi_hip.

    WanDiT.from_state_dict(sd, **cfg)      <- WanModel(**cfg).load_state_dict(sd)   models/wan_video_dit.py:407-470
    WanDiT.from_module(wan_model)          <- an existing reference WanModel (weights borrowed, not copied)
    model_fn_wan_video(dit, x, timestep, context, clip_feature, y, ...)             pipelines/svi_video.py:74-137
    WanDiT.block_forward(i, x, context, t_mod, grid)   <- dit.blocks[i](x, context, t_mod, freqs)    dit:354-374

The handle borrows the device pointers of the bound tensors; this object keeps those tensors alive and
re-binds them on `rebind()` (call it after a LoRA merge or any operation that moves parameter storage).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import _lib as L


def _version(t: torch.Tensor) -> int:
    """Version counter of a tensor (bumped by every in-place write); inference tensors do not keep one."""
    return 0 if t.is_inference() else t._version


def config_of_module(wan_model) -> dict:
    """WanDiT's constructor arguments read off a reference `WanModel` instance (models/wan_video_dit.py:407-470): only attributes the
    real class sets in its constructor — `dim`, `freq_dim`, `has_image_input`, `patch_embedding` (nn.Conv3d), `text_embedding[0]`
    (nn.Linear), `blocks[i].{ffn_dim, num_heads, norm1}` (DiTBlock :321-336), `head.head` (Head :392-399), `enable_multitalk`.
    tests/test_reference_keys.py runs it on the real class (meta device) for the reference converter's four constructor tables."""
    blk = wan_model.blocks[0]
    pe = wan_model.patch_embedding
    kt, kh, kw = (int(k) for k in pe.kernel_size)
    return dict(dim=int(wan_model.dim), in_dim=int(pe.in_channels), ffn_dim=int(blk.ffn_dim),
                out_dim=int(wan_model.head.head.out_features) // (kt * kh * kw), text_dim=int(wan_model.text_embedding[0].in_features),
                freq_dim=int(wan_model.freq_dim), eps=float(blk.norm1.eps), patch_size=(kt, kh, kw), num_heads=int(blk.num_heads),
                num_layers=len(wan_model.blocks), has_image_input=bool(wan_model.has_image_input),
                enable_multitalk=bool(getattr(wan_model, "enable_multitalk", False)))


class PromptPins:
    """Bookkeeping that makes the pointer-keyed context cache of the C side (svi_dit_context_cache) safe to leave on.

    The C side recognises a prompt embedding by its device pointer.  Two things can make a pointer lie: the tensor is freed and
    the allocator hands its storage to the next prompt's embedding (same shape, same address), or the tensor is written in
    place.  So every tensor the cache may have seen is kept alive here — its storage cannot be recycled while it is pinned —
    together with its version counter and shape; `admit` says when the C-side entries must be dropped."""

    def __init__(self, capacity: int = 8):
        self.capacity = capacity
        self._pins: Dict[int, Tuple[torch.Tensor, int, Tuple[int, ...]]] = {}

    def __len__(self) -> int:
        return len(self._pins)

    def clear(self) -> None:
        self._pins = {}

    def repin(self, tensors) -> None:
        """The C-side entries of these tensors were just recomputed from their present contents (WanDiT.refill_context): record the versions."""
        for t in tensors:
            if t is not None:
                self._pins[t.data_ptr()] = (t, _version(t), tuple(t.shape))

    def admit(self, tensors) -> bool:
        """Pin the prompt-side tensors of one call.  Returns True when the cache entries made so far are no longer trustworthy
        (a pinned tensor changed, or more prompts are in flight than are kept) — the caller then drops them; the tensors of this
        call are pinned either way."""
        tensors = [t for t in tensors if t is not None]
        stale = any(p is not None and (p[1] != _version(t) or p[2] != tuple(t.shape))
                    for t in tensors for p in [self._pins.get(t.data_ptr())])
        fresh = {t.data_ptr() for t in tensors if t.data_ptr() not in self._pins}
        drop = stale or len(self._pins) + len(fresh) > self.capacity
        if drop:
            self._pins = {}
        for t in tensors:
            self._pins[t.data_ptr()] = (t, _version(t), tuple(t.shape))
        return drop


class WanDiT:
    def __init__(self, dim: int, in_dim: int, ffn_dim: int, out_dim: int, text_dim: int, freq_dim: int, eps: float,
                 patch_size: Tuple[int, int, int], num_heads: int, num_layers: int, has_image_input: bool, enable_multitalk: bool = False):
        self.dim, self.in_dim, self.ffn_dim, self.out_dim = dim, in_dim, ffn_dim, out_dim
        self.text_dim, self.freq_dim, self.eps, self.patch_size = text_dim, freq_dim, eps, tuple(patch_size)
        self.num_heads, self.num_layers, self.has_image_input = num_heads, num_layers, bool(has_image_input)
        self.enable_multitalk = bool(enable_multitalk)
        self._audio = None                  # the armed audio windows (kept alive: the C side borrows their pointers)
        cfg = L.DitConfig(dim, in_dim, ffn_dim, out_dim, text_dim, freq_dim, eps, *self.patch_size, num_heads,
                          num_layers, int(self.has_image_input), int(self.enable_multitalk))
        h = C.c_void_p()
        L.check(L.lib().svi_dit_create(C.byref(cfg), C.byref(h)), "svi_dit_create")
        self._h = h
        self._params: Dict[str, torch.Tensor] = {}
        self._fp8_sources: Dict[str, torch.Tensor] = {}        # fp8-stored parameters whose bf16 copies are bound (FP8 storage mode)
        self._param_versions = []
        self._epoch = 0                     # host-side: moves on bind / rebind / context_cache(); part of a captured graph's key
        self._ffn_mx8 = False               # opt-in MX-fp8 MLP (ffn_fp8_mfma)
        self._proj_mx8 = False              # opt-in MX-fp8 q / k / v / o and cross-attention q / o projections (proj_fp8_mfma)
        self._ctx_cache_on = False
        self._ctx_pins = PromptPins()

    # ---- construction -------------------------------------------------------------------------
    @classmethod
    def from_state_dict(cls, state_dict: Dict[str, torch.Tensor], device="cuda", **cfg) -> "WanDiT":
        m = cls(**cfg)
        m.bind({k: (v.to(device=device) if v.dtype == torch.float8_e4m3fn else v.to(device=device, dtype=torch.bfloat16)).contiguous()
                for k, v in state_dict.items()})
        return m

    @classmethod
    def from_module(cls, wan_model) -> "WanDiT":
        """Borrow the parameters of a reference `WanModel` (already on the GPU, bf16)."""
        m = cls(**config_of_module(wan_model))
        m.bind(dict(wan_model.state_dict()))
        return m

    def check_module_in_place(self, wan_model) -> None:
        """The parameters of the nn.Module this handle borrows from must still be where they were bound: on the GPU, same storage.  A parameter
        that moved ON the device (re-created by a loader, `.to()` to another GPU dtype and back) is re-bound; one that left the device, or a
        module tree whose keys changed (VRAM-management wrappers rename them: `norm3.module.weight`), is refused — the kernels would
        otherwise keep reading the stale copies this handle holds alive."""
        sd = wan_model.state_dict()
        moved = False
        for name, p in sd.items():
            bound = self._fp8_sources.get(name)
            if bound is None:
                bound = self._params.get(name)
            if bound is None:
                raise RuntimeError(f"svi_hip: pipe.dit's state dict now has the key {name!r}, which was not there at install() — modules were wrapped or "
                                   "replaced (enable_vram_management?): call svi_hip.install(pipe) again")
            if not p.is_cuda:
                raise RuntimeError(f"svi_hip: parameter {name} of pipe.dit now lives on {p.device} — the DiT was offloaded after install(); the HIP backend "
                                   "keeps it resident: call svi_hip.install(pipe) again (it moves the models back and neutralises the offload calls)")
            if p.data_ptr() != bound.data_ptr() or p.dtype != bound.dtype:
                moved = True
        if len(sd) != len(self._params):
            raise RuntimeError("svi_hip: pipe.dit lost parameters since install(): call svi_hip.install(pipe) again")
        if moved:
            self._fp8_sources = {}
            self.bind({k: v for k, v in sd.items()})

    def bind(self, state_dict: Dict[str, torch.Tensor]) -> None:
        lib = L.lib()
        for name, t in state_dict.items():
            if t.is_cuda and t.dtype == torch.float8_e4m3fn:
                # the reference's FP8 mode (test_svi.py:337): parameters live as e4m3 and are cast to bf16 in front of every use; the
                # cast is exact, so it is done once here and the bf16 copy is what the kernels read (ops.fp8_e4m3_to_bf16)
                from .ops import fp8_e4m3_to_bf16
                self._fp8_sources[name] = t
                t = fp8_e4m3_to_bf16(t)
            if not t.is_cuda or t.dtype != torch.bfloat16 or not t.is_contiguous():
                raise RuntimeError(f"parameter {name} must be a contiguous CUDA bf16 tensor (got {t.device}, {t.dtype})")
            shape = (C.c_int64 * t.dim())(*t.shape)
            L.check(lib.svi_dit_bind_weight(self._h, name.encode(), t.data_ptr(), L.SVI_BF16, shape, t.dim()),
                    f"bind {name}")
            self._params[name] = t
        L.check(lib.svi_dit_check_bound(self._h), "svi_dit_check_bound")
        # what weights_changed() watches: the bound bf16 tensors and, in FP8 storage mode, the e4m3 tensors they were cast from
        self._param_versions = [(t, _version(t)) for t in self._params.values()] + [(t, _version(t)) for t in self._fp8_sources.values()]
        self._epoch += 1
        if self._ffn_mx8:
            self.ffn_fp8_mfma(True)         # the e4m3 tensors may have moved with the re-bind
        if self._proj_mx8:
            self.proj_fp8_mfma(True)

    def rebind(self) -> None:
        """Re-read every parameter's address (after a LoRA merge, .to(), an offload round trip ...); drops the context cache.
        FP8-stored parameters are cast again from their e4m3 sources (an in-place update of a source is picked up)."""
        self.bind({**self._params, **self._fp8_sources})

    def ffn_fp8_mfma(self, enable: bool = True) -> None:
        """Opt-in: run both MLP GEMMs of every block on the MX block-scaled fp8 matrix path (svi_gemm_mx8; ~2x the bf16 MFMA rate).
        Needs the FP8 weight STORAGE mode (parameters bound as float8_e4m3fn, the reference's test_svi.py:337 mode): the stored e4m3
        bytes of ffn.0 / ffn.2 are used as they are (unit scales), activations are quantised per row and 32-element block in front of
        each GEMM.  The reference performs no fp8 arithmetic — this mode has its own tolerance (tests/test_gpu_mx8.py) and bench line
        (bench.py --fp8-mfma) and is never a default."""
        lib = L.lib()
        if enable:
            for l in range(self.num_layers):
                for which in (0, 2):
                    name = f"blocks.{l}.ffn.{which}.weight"
                    t = self._fp8_sources.get(name)
                    if t is None:
                        raise RuntimeError(f"ffn_fp8_mfma needs {name} bound as float8_e4m3fn (FP8 weight storage mode)")
                    L.check(lib.svi_dit_bind_ffn_fp8(self._h, l, which, t.data_ptr()), f"svi_dit_bind_ffn_fp8 {name}")
        L.check(lib.svi_dit_ffn_mx8(self._h, 1 if enable else 0), "svi_dit_ffn_mx8")
        self._ffn_mx8 = bool(enable)
        self._epoch += 1

    _PROJ_FP8 = ("self_attn.q", "self_attn.k", "self_attn.v", "self_attn.o", "cross_attn.q", "cross_attn.o")      # svi_dit_bind_ffn_fp8 which = 10 + index

    def proj_fp8_mfma(self, enable: bool = True) -> None:
        """Opt-in, with ffn_fp8_mfma's caveats (NOT the reference's arithmetic; own tolerance, tests/test_gpu_mx8.py; bench.py --fp8-all): the block's other six
        projections — self-attention q, k, v, o and cross-attention q, o — on the MX block-scaled fp8 matrix path (svi_dit_proj_mx8).  Needs the FP8 weight
        STORAGE mode: the stored e4m3 bytes are used as they are (unit scales); the LayerNorm output is quantised once for q, k and V^T, each attention output once
        for its output projection.  The prompt-side K / V projections (context cache) and sequence-parallel shards keep bf16."""
        lib = L.lib()
        if enable:
            for l in range(self.num_layers):
                for i, mod in enumerate(self._PROJ_FP8):
                    name = f"blocks.{l}.{mod}.weight"
                    t = self._fp8_sources.get(name)
                    if t is None:
                        raise RuntimeError(f"proj_fp8_mfma needs {name} bound as float8_e4m3fn (FP8 weight storage mode)")
                    L.check(lib.svi_dit_bind_ffn_fp8(self._h, l, 10 + i, t.data_ptr()), f"svi_dit_bind_ffn_fp8 {name}")
        L.check(lib.svi_dit_proj_mx8(self._h, 1 if enable else 0), "svi_dit_proj_mx8")
        self._proj_mx8 = bool(enable)
        self._epoch += 1

    def epoch(self) -> int:
        return self._epoch

    def generation(self) -> int:
        """svi_dit_generation: moves when device state a captured hipGraph relies on stops being valid."""
        return int(L.lib().svi_dit_generation(self._h))

    def weights_changed(self) -> bool:
        """True when a bound parameter was written in place since bind() (merge_lora_, load_state_dict into the same storage):
        the cached cross-attention K / V were projected with the old values."""
        return any(_version(t) != v for t, v in self._param_versions)

    def _refresh_if_weights_changed(self) -> None:
        """Bound bf16 tensors are read in place by the kernels: an in-place write to one needs a re-bind only for what was DERIVED from it —
        the cached prompt-side projections (context cache on).  In FP8 storage mode the bound bf16 tensors themselves are derived (cast
        from the e4m3 sources at bind): an in-place write to a source re-binds whether or not the cache is on (otherwise attention
        would keep the stale casts while the MX-fp8 MLP reads the new bytes)."""
        fp8_stale = any(_version(t) != v for t, v in self._param_versions if t.dtype == torch.float8_e4m3fn)
        if fp8_stale or (self._ctx_cache_on and self.weights_changed()):
            self.rebind()
            self._ctx_pins.clear()

    def context_cache(self, enable: bool) -> None:
        """Reuse the projected context and the cross-attention K / V^T of every block across forwards that are handed the
        same context tensor (same storage, unchanged contents); see svi_dit_context_cache in include/svi_hip.h.  While it is on,
        context / clip_feature must be contiguous CUDA bf16 tensors (a converted temporary would be recognised by an address the
        allocator recycles); they are kept alive here and an in-place write to one of them drops the cache (PromptPins)."""
        L.check(L.lib().svi_dit_context_cache(self._h, 1 if enable else 0), "svi_dit_context_cache")
        self._ctx_cache_on = bool(enable)
        self._ctx_pins.clear()
        self._epoch += 1

    def refill_context(self, context: torch.Tensor, clip_feature: Optional[torch.Tensor] = None) -> None:
        """The rolling window's clip boundary: `context` (and `clip_feature`) are tensors the context cache already knows, and the caller has
        just written the NEXT clip's prompt embedding (CLIP feature) into them in place.  The cache entry keyed by their addresses is recomputed
        in the buffers it owns (svi_dit_context_refill: projected context, every block's cross-attention K / V^T); nothing moves, so a captured
        step graph that reads those buffers stays valid (epoch() and generation() do not change) and the pins take the tensors' new versions."""
        if not self._ctx_cache_on:
            raise RuntimeError("refill_context: the context cache is off")
        if not self.has_image_input:
            clip_feature = None
        for t in (context, clip_feature):
            if t is not None and (not t.is_cuda or t.dtype != torch.bfloat16 or not t.is_contiguous()):
                raise ValueError("refill_context: context / clip_feature must be contiguous CUDA bf16 tensors")
        if context.dim() != 3 or context.shape[2] != self.text_dim:
            raise ValueError(f"refill_context: context must be [B, Lc, {self.text_dim}] (got {tuple(context.shape)})")
        self._refresh_if_weights_changed()
        B, Lc = context.shape[0], context.shape[1]
        for b in range(B):          # the entries are keyed per sample (svi_dit_forward walks the batch with these offsets)
            cp = context.data_ptr() + b * Lc * self.text_dim * 2
            fp = None if clip_feature is None else clip_feature.data_ptr() + b * 257 * 1280 * 2
            L.check(L.lib().svi_dit_context_refill(self._h, cp, fp, Lc, L.current_stream()), "svi_dit_context_refill")
        self._ctx_pins.repin([context, clip_feature])

    def _prompt_args(self, *tensors):
        """The prompt-side inputs (context(s), clip_feature) as contiguous bf16; see context_cache()."""
        self._refresh_if_weights_changed()
        if not self._ctx_cache_on:
            return tuple(None if t is None else t.to(torch.bfloat16).contiguous() for t in tensors)
        for t in tensors:
            if t is not None and (not t.is_cuda or t.dtype != torch.bfloat16 or not t.is_contiguous()):
                raise ValueError("with the context cache on, context / clip_feature must be contiguous CUDA bf16 tensors that live "
                                 "across the calls (convert once, outside the step loop)")
        if self._ctx_pins.admit(tensors):
            L.check(L.lib().svi_dit_context_cache(self._h, 1), "svi_dit_context_cache")      # re-enabling drops every entry
            self._epoch += 1
        return tensors

    def check_inputs(self, x, contexts, clip_feature=None, y=None, add_condition=None) -> None:
        """Shape contract of model_fn_wan_video's inputs.  The C side walks raw pointers with the sizes it is told, so what the
        reference would reject with a shape error (wrong channel counts, a missing y / clip_feature, a foreign text width) is
        rejected here instead of being read out of bounds."""
        if x.dim() != 5 or x.shape[1] != 16:
            raise ValueError(f"latents must be [B, 16, T, H, W] (got {tuple(x.shape)})")
        B, _, T, H, W = x.shape
        pt, ph, pw = self.patch_size
        if T % pt or H % ph or W % pw:
            raise ValueError(f"latent grid {T}x{H}x{W} is not divisible by the patch size {self.patch_size}")
        for c in contexts:
            if c is None or c.dim() != 3 or c.shape[0] != B or c.shape[2] != self.text_dim:
                raise ValueError(f"context must be [B={B}, Lc, text_dim={self.text_dim}] (got {None if c is None else tuple(c.shape)})")
        if self.in_dim > 16:
            if y is None or tuple(y.shape) != (B, self.in_dim - 16, T, H, W):
                raise ValueError(f"this model takes y of shape {(B, self.in_dim - 16, T, H, W)} (got {None if y is None else tuple(y.shape)})")
        elif y is not None:
            raise ValueError("y was given but the model has in_dim == 16")
        if self.has_image_input:
            if clip_feature is None or tuple(clip_feature.shape) != (B, 257, 1280):
                raise ValueError(f"has_image_input model needs clip_feature [B={B}, 257, 1280] (got "
                                 f"{None if clip_feature is None else tuple(clip_feature.shape)})")
        if add_condition is not None and tuple(add_condition.shape) != (B, self.tokens(T, H, W), self.dim):
            raise ValueError(f"add_condition must be [B, L, dim] = {(B, self.tokens(T, H, W), self.dim)} (got {tuple(add_condition.shape)})")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                L.lib().svi_dit_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- forward ------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, timestep: torch.Tensor, context: torch.Tensor,
                clip_feature: Optional[torch.Tensor] = None, y: Optional[torch.Tensor] = None,
                add_condition: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                tea_mode: int = 0, residual: Optional[torch.Tensor] = None, audio_embed_tuple=None) -> torch.Tensor:
        """tea_mode / residual: TeaCache plumbing (svi_dit_forward_tea): 1 = also write residual [B, L, dim] = x_after_blocks -
        x_before_blocks, 2 = skip the blocks and add `residual` instead.  audio_embed_tuple: the talk variant (set_audio) for this call."""
        if not x.is_cuda:
            raise RuntimeError("svi_hip runs on the GPU only")
        if audio_embed_tuple is not None:       # given for this call: arm it (None leaves whatever set_audio() armed in place)
            same = self._audio is not None and all(a.data_ptr() == b.data_ptr() and a.dtype == b.dtype for a, b in zip(self._audio, audio_embed_tuple))
            if not same:
                self.set_audio(audio_embed_tuple)
        if self._audio is not None and self._audio[1].shape[1] != x.shape[2] - 1:
            raise ValueError(f"audio windows cover {self._audio[1].shape[1] + 1} latent frames, the latents have {x.shape[2]}")
        self.check_inputs(x, (context,), clip_feature, y, add_condition)
        if not self.has_image_input:
            clip_feature = None                    # the reference ignores it without an image branch (svi_video.py:96-99)
        x = x.to(torch.bfloat16).contiguous()
        context, clip_feature = self._prompt_args(context, clip_feature)
        timestep = timestep.to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
        B, _, T, H, W = x.shape
        if timestep.numel() != B:
            timestep = timestep.expand(B).contiguous()
        if y is not None:
            y = y.to(torch.bfloat16).contiguous()
        if add_condition is not None:
            add_condition = add_condition.to(torch.bfloat16).contiguous()
        if out is None:
            out = torch.empty((B, self.out_dim, T, H, W), dtype=torch.bfloat16, device=x.device)
        if tea_mode:
            if residual is None or not residual.is_cuda or residual.dtype != torch.bfloat16 or not residual.is_contiguous() \
                    or residual.numel() != B * self.tokens(T, H, W) * self.dim:
                raise ValueError("tea_mode needs a contiguous CUDA bf16 residual of shape [B, L, dim]")
            L.check(L.lib().svi_dit_forward_tea(self._h, L.ptr(x), L.ptr(timestep), L.ptr(context), L.ptr(clip_feature), L.ptr(y),
                                                L.ptr(add_condition), L.ptr(out), B, T, H, W, context.shape[1], int(tea_mode),
                                                L.ptr(residual), L.current_stream()), "svi_dit_forward_tea")
            return out
        L.check(L.lib().svi_dit_forward(self._h, L.ptr(x), L.ptr(timestep), L.ptr(context), L.ptr(clip_feature), L.ptr(y),
                                        L.ptr(add_condition), L.ptr(out), B, T, H, W, context.shape[1],
                                        L.current_stream()), "svi_dit_forward")
        return out

    def set_audio(self, audio_embed_tuple) -> None:
        """Arm (or, with None, disarm) the talk variant for the following forward() calls: audio_embed_tuple = (first [1, 1, 5, 12, 768],
        latter [1, T-1, 8, 12, 768]) as SVITalkVideoPipeline builds it (svi_video_talk.py:425-444; model_fn_wan_talk_video :126-127)."""
        if audio_embed_tuple is None:
            L.check(L.lib().svi_dit_set_audio(self._h, None, None, 0), "svi_dit_set_audio")
            self._audio = None
            return
        if not self.enable_multitalk:
            raise ValueError("audio_embed_tuple was given to a model without enable_multitalk")
        a0, a1 = audio_embed_tuple
        if tuple(a0.shape) != (1, 1, 5, 12, 768) or a1.dim() != 5 or tuple(a1.shape[:1] + a1.shape[2:]) != (1, 8, 12, 768):
            raise ValueError(f"audio_embed_tuple must be ([1,1,5,12,768], [1,T-1,8,12,768]) (got {tuple(a0.shape)}, {tuple(a1.shape)})")
        a0 = a0.to(device="cuda", dtype=torch.bfloat16).contiguous()
        a1 = a1.to(device="cuda", dtype=torch.bfloat16).contiguous()
        L.check(L.lib().svi_dit_set_audio(self._h, L.ptr(a0), L.ptr(a1) if a1.shape[1] else None, a1.shape[1]), "svi_dit_set_audio")
        self._audio = (a0, a1)

    def tokens(self, T: int, H: int, W: int) -> int:
        pt, ph, pw = self.patch_size
        return (T // pt) * (H // ph) * (W // pw)

    def time_mod(self, timestep: torch.Tensor) -> torch.Tensor:
        """t_mod bf16 [B, 6, dim] = time_projection(silu(time_embedding(sinusoidal(t)))) (svi_video.py:92-93)."""
        timestep = timestep.to(device="cuda", dtype=torch.float32).reshape(-1).contiguous()
        out = torch.empty((timestep.numel(), 6, self.dim), dtype=torch.bfloat16, device="cuda")
        L.check(L.lib().svi_dit_time_mod(self._h, L.ptr(timestep), L.ptr(out), timestep.numel(), L.current_stream()), "svi_dit_time_mod")
        return out

    __call__ = forward

    def forward_cfg_pair(self, x: torch.Tensor, timestep: torch.Tensor, context_cond: torch.Tensor, context_uncond: torch.Tensor,
                         clip_feature: Optional[torch.Tensor] = None, y: Optional[torch.Tensor] = None,
                         add_condition: Optional[torch.Tensor] = None, out_cond: Optional[torch.Tensor] = None,
                         out_uncond: Optional[torch.Tensor] = None):
        """Both forwards of a CFG step (svi_video.py:401-408) in one call: what precedes the first use of the prompt (timestep
        embedding, patchify, block 0's self-attention) is computed once; outputs are bit-identical to two forward() calls."""
        if not x.is_cuda:
            raise RuntimeError("svi_hip runs on the GPU only")
        if context_cond.shape != context_uncond.shape:
            raise ValueError("the two prompt embeddings of a CFG pair must have the same shape")
        self.check_inputs(x, (context_cond, context_uncond), clip_feature, y, add_condition)
        if not self.has_image_input:
            clip_feature = None
        x = x.to(torch.bfloat16).contiguous()
        context_cond, context_uncond, clip_feature = self._prompt_args(context_cond, context_uncond, clip_feature)
        timestep = timestep.to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
        B, _, T, H, W = x.shape
        if timestep.numel() != B:
            timestep = timestep.expand(B).contiguous()
        if y is not None:
            y = y.to(torch.bfloat16).contiguous()
        if add_condition is not None:
            add_condition = add_condition.to(torch.bfloat16).contiguous()
        if out_cond is None:
            out_cond = torch.empty((B, self.out_dim, T, H, W), dtype=torch.bfloat16, device=x.device)
        if out_uncond is None:
            out_uncond = torch.empty((B, self.out_dim, T, H, W), dtype=torch.bfloat16, device=x.device)
        L.check(L.lib().svi_dit_forward_cfg_pair(self._h, L.ptr(x), L.ptr(timestep), L.ptr(context_cond), L.ptr(context_uncond),
                                                 L.ptr(clip_feature), L.ptr(y), L.ptr(add_condition), L.ptr(out_cond), L.ptr(out_uncond),
                                                 B, T, H, W, context_cond.shape[1], L.current_stream()), "svi_dit_forward_cfg_pair")
        return out_cond, out_uncond

    def block_forward(self, layer: int, x: torch.Tensor, context: torch.Tensor, t_mod: torch.Tensor,
                      grid: Tuple[int, int, int]) -> torch.Tensor:
        """x [1, L, dim] (not modified; a copy is updated and returned), context [1, Lc(+257), dim] already
        projected, t_mod [1, 6, dim]."""
        f, h, w = grid
        xo = x.to(torch.bfloat16).contiguous().clone()
        context = context.to(torch.bfloat16).contiguous()
        t_mod = t_mod.to(torch.bfloat16).contiguous()
        lc = context.shape[-2] - (257 if self.has_image_input else 0)
        L.check(L.lib().svi_dit_block_forward(self._h, layer, L.ptr(xo), L.ptr(context), L.ptr(t_mod), f, h, w, lc,
                                              L.current_stream()), "svi_dit_block_forward")
        return xo


def model_fn_wan_video(dit: WanDiT, x: torch.Tensor, timestep: torch.Tensor, context: torch.Tensor,
                       clip_feature: Optional[torch.Tensor] = None, y: Optional[torch.Tensor] = None,
                       tea_cache=None, add_condition=None, use_unified_sequence_parallel: bool = False,
                       **kwargs) -> torch.Tensor:
    """Same signature as pipelines/svi_video.py:74-85.

    tea_cache: the reference's TeaCache object (svi_video.py:23-72) or svi_hip.TeaCache — both are driven through their own
    check(dit, x, t_mod): the skip decision is theirs (host arithmetic on t_mod), the residual bookkeeping is done on the device
    (`previous_residual` holds our [B, L, dim] buffer).
    use_unified_sequence_parallel: with an initialised process group of more than one rank (dit.sp_group, default: the world) the
    forward is spread Ulysses-style over the ranks (svi_hip/sequence_parallel.py); with one rank it is the plain forward, as in
    the reference.  The two combine as in the reference (svi_video.py:112-131): the skip decision is taken on t_mod (identical on every
    rank), the residual a rank stores / adds covers its own token rows."""
    fwd, rows = dit.forward, None
    if use_unified_sequence_parallel:
        import torch.distributed as dist
        grp = getattr(dit, "sp_group", None)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(grp) > 1:   # svi_video.py:119-121
            from .sequence_parallel import forward_distributed
            fwd = lambda *a, **kw: forward_distributed(dit, *a, group=grp, **kw)      # noqa: E731
            B, _, T, H, W = x.shape
            rows = dit.tokens(T, H, W) // dist.get_world_size(grp)                     # TeaCache residuals hold this rank's rows only
    if tea_cache is None:
        return fwd(x, timestep, context, clip_feature=clip_feature, y=y, add_condition=add_condition)
    t_mod = dit.time_mod(timestep)
    # check() clones `x` only to form the residual later (store()); the residual is formed on the device here, so a token suffices
    skip = tea_cache.check(dit, t_mod[:, :1, :1], t_mod)
    B, _, T, H, W = x.shape
    shape = (B, dit.tokens(T, H, W), dit.dim) if rows is None else (rows, dit.dim)
    if skip:
        res = tea_cache.previous_residual
        if res is None:
            raise RuntimeError("TeaCache asked to skip before any residual was stored")
        return fwd(x, timestep, context, clip_feature=clip_feature, y=y, add_condition=add_condition, tea_mode=2, residual=res)
    res = getattr(tea_cache, "_svi_residual", None)
    if res is None or tuple(res.shape) != shape:
        res = torch.empty(shape, dtype=torch.bfloat16, device=x.device)
        tea_cache._svi_residual = res
    out = fwd(x, timestep, context, clip_feature=clip_feature, y=y, add_condition=add_condition, tea_mode=1, residual=res)
    tea_cache.previous_residual = res            # what TeaCache.store() would have computed (svi_video.py:64-66)
    tea_cache.previous_hidden_states = None
    return out


def model_fn_wan_talk_video(dit: WanDiT, x: torch.Tensor, timestep: torch.Tensor, context: torch.Tensor,
                            clip_feature: Optional[torch.Tensor] = None, y: Optional[torch.Tensor] = None, tea_cache=None,
                            add_condition=None, audio_embed_tuple=None, use_unified_sequence_parallel: bool = False,
                            use_controlnet: bool = False, **kwargs) -> torch.Tensor:
    """Same signature as pipelines/svi_video_talk.py:83-96.  The audio windows arm the per-block audio cross-attention for this call
    (WanDiT.set_audio); everything else is model_fn_wan_video.  Not served: use_controlnet (blocks that also return add_condition) and
    sequence parallelism together with audio."""
    if use_controlnet:
        raise NotImplementedError("the talk variant's controlnet blocks are not served by the HIP backend")
    if audio_embed_tuple is None:
        raise ValueError("model_fn_wan_talk_video needs audio_embed_tuple (the reference dereferences it unconditionally, svi_video_talk.py:126)")
    if use_unified_sequence_parallel:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(getattr(dit, "sp_group", None)) > 1:
            raise NotImplementedError("the talk variant's per-frame audio attention is not served on sequence shards")
    dit.set_audio(audio_embed_tuple)
    try:
        return model_fn_wan_video(dit, x, timestep, context, clip_feature=clip_feature, y=y, tea_cache=tea_cache, add_condition=add_condition)
    finally:
        dit.set_audio(None)
