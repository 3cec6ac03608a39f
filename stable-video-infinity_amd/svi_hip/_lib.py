"""ctypes binding of libsvi_hip.so (declared in include/svi_hip.h).

There is no fallback: if the shared object is missing or a call fails, a RuntimeError carrying
svi_last_error() is raised.  Nothing here touches oracle/.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVI_HIP_LIB") or os.path.join(_HERE, "libsvi_hip.so")   # SVI_HIP_LIB: A/B a variant build (tools/build_variant.py)

SVI_OK = 0
SVI_BF16, SVI_F32 = 0, 1
EPI_BIAS, EPI_BIAS_GELU_TANH, EPI_BIAS_GATE_RES, EPI_BIAS_GELU_ERF, EPI_BIAS_SILU, EPI_BIAS_RELU = 0, 1, 2, 3, 4, 5

# every symbol include/svi_hip.h declares: (name, restype, argtypes)
_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class DitConfig(C.Structure):
    _fields_ = [("dim", _i32), ("in_dim", _i32), ("ffn_dim", _i32), ("out_dim", _i32), ("text_dim", _i32),
                ("freq_dim", _i32), ("eps", _f32), ("patch_t", _i32), ("patch_h", _i32), ("patch_w", _i32),
                ("num_heads", _i32), ("num_layers", _i32), ("has_image_input", _i32), ("enable_multitalk", _i32)]


class T5Config(C.Structure):
    _fields_ = [("vocab", _i32), ("dim", _i32), ("dim_attn", _i32), ("dim_ffn", _i32), ("num_heads", _i32), ("num_layers", _i32),
                ("num_buckets", _i32), ("max_dist", _i32), ("shared_pos", _i32)]


class ClipConfig(C.Structure):
    _fields_ = [("image_size", _i32), ("patch_size", _i32), ("dim", _i32), ("mlp_ratio", _i32), ("num_heads", _i32), ("num_layers", _i32),
                ("layers_used", _i32), ("norm_eps", _f32)]


SYMBOLS = [
    ("svi_last_error", C.c_char_p, []),
    ("svi_abi_version", _i32, []),
    ("svi_device_count", _i32, []),
    ("svi_switches_reload", _i32, []),
    ("svi_switch_state", _i32, [C.c_char_p]),
    ("svi_dit_create", _i32, [C.POINTER(DitConfig), C.POINTER(_vp)]),
    ("svi_dit_destroy", _i32, [_vp]),
    ("svi_dit_bind_weight", _i32, [_vp, C.c_char_p, _vp, _i32, C.POINTER(_i64), _i32]),
    ("svi_dit_check_bound", _i32, [_vp]),
    ("svi_dit_forward", _i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    ("svi_dit_sp_begin", _i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    ("svi_dit_sp_begin_pair", _i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    ("svi_dit_sp_block_qkv", _i32, [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    ("svi_dit_sp_block_qkv_part", _i32, [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    ("svi_sp_unpack_vt", _i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    ("svi_sp_unpack_out", _i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    ("svi_dit_sp_block_rest", _i32, [_vp, _i32, _vp, _vp]),
    ("svi_dit_sp_tea", _i32, [_vp, _i32, _vp, _vp]),
    ("svi_dit_sp_head", _i32, [_vp, _vp, _vp]),
    ("svi_dit_unpatchify", _i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    ("svi_dit_head_ld", _i32, [_vp]),
    ("svi_dit_generation", _i64, [_vp]),
    ("svi_stream_buffers_release", _i32, [_vp, _i32]),
    ("svi_attention_last_flagged", _i32, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    ("svi_attention_vt_fwd", _i32, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    ("svi_dit_set_audio", _i32, [_vp, _vp, _vp, _i32]),
    ("svi_cfg3_step", _i32, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _vp]),
    ("svi_dit_time_mod", _i32, [_vp, _vp, _vp, _i32, _vp]),
    ("svi_dit_forward_tea", _i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    ("svi_dit_forward_cfg_pair", _i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    ("svi_dit_context_cache", _i32, [_vp, _i32]),
    ("svi_dit_context_refill", _i32, [_vp, _vp, _vp, _i32, _vp]),
    ("svi_linear_row_stats", _i32, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _f32, _vp, _i32, _vp, _vp]),
    ("svi_cross_attention_fwd", _i32, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _f32, _vp]),
    ("svi_dit_block_forward", _i32, [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    ("svi_attention_fwd", _i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    ("svi_layernorm_modulate", _i32, [_vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp]),
    ("svi_rmsnorm_rope", _i32, [_vp, _i32, _i32, _i32, _vp, _f32, _i32, _i32, _i32, _i32, _i32, _vp]),
    ("svi_gemm_bf16", _i32, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp, _vp, _i32, _vp]),
    ("svi_cfg_step", _i32, [_vp, _vp, _vp, _i64, _f32, _f32, _vp]),
    ("svi_mx8_quantize", _i32, [_vp, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp]),
    ("svi_gemm_mx8", _i32, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _vp]),
    ("svi_dit_bind_ffn_fp8", _i32, [_vp, _i32, _i32, _vp]),
    ("svi_dit_ffn_mx8", _i32, [_vp, _i32]),
    ("svi_gemm_mx8_wscaled", _i32, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    ("svi_dit_proj_mx8", _i32, [_vp, _i32]),
    ("svi_fp8_e4m3_to_bf16", _i32, [_vp, _vp, _i64, _vp]),
    ("svi_gemm_plan", _i32, [_i32, _i32, _i32, _i32, _i32, _i32, C.POINTER(_i32)]),
    ("svi_attention_plan", _i32, [_i32, _i32, _i32, _i32, C.POINTER(_i32)]),
    ("svi_prof_enable", _i32, [_i32]),
    ("svi_prof_summary", _i32, [C.c_char_p, _i64]),
    ("svi_prof_select", _i32, [C.c_char_p]),
    ("svi_vae_create", _i32, [C.POINTER(_vp)]),
    ("svi_vae_destroy", _i32, [_vp]),
    ("svi_vae_bind_weight", _i32, [_vp, C.c_char_p, _vp, _i32, C.POINTER(_i64), _i32]),
    ("svi_vae_check_bound", _i32, [_vp]),
    ("svi_video_to_u8", _i32, [_vp, _vp, _i32, _i32, _i32, _vp]),
    ("svi_u8_to_video", _i32, [_vp, _vp, _i32, _i32, _i32, _vp]),
    ("svi_vae_decode", _i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    ("svi_vae_encode", _i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    ("svi_vae_tiled_decode", _i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    ("svi_vae_tiled_encode", _i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    ("svi_pose_create", _i32, [_i32, _i32, C.POINTER(_vp)]),
    ("svi_pose_destroy", _i32, [_vp]),
    ("svi_pose_bind_weight", _i32, [_vp, C.c_char_p, _vp, _i32, C.POINTER(_i64), _i32]),
    ("svi_pose_check_bound", _i32, [_vp]),
    ("svi_pose_tokens", _i32, [_vp, _i32, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    ("svi_pose_forward", _i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    ("svi_t5_create", _i32, [C.POINTER(T5Config), C.POINTER(_vp)]),
    ("svi_t5_destroy", _i32, [_vp]),
    ("svi_t5_bind_weight", _i32, [_vp, C.c_char_p, _vp, _i32, C.POINTER(_i64), _i32]),
    ("svi_t5_check_bound", _i32, [_vp]),
    ("svi_t5_forward", _i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    ("svi_t5_relative_buckets", _i32, [_i32, _i32, _i32, C.POINTER(_i32)]),
    ("svi_t5_device_buckets", _i32, [_vp, _i32, C.POINTER(_i32)]),
    ("svi_clip_create", _i32, [C.POINTER(ClipConfig), C.POINTER(_vp)]),
    ("svi_clip_destroy", _i32, [_vp]),
    ("svi_clip_bind_weight", _i32, [_vp, C.c_char_p, _vp, _i32, C.POINTER(_i64), _i32]),
    ("svi_clip_check_bound", _i32, [_vp]),
    ("svi_clip_tokens", _i32, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    ("svi_clip_encode_image", _i32, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
]

_lib = None


def lib() -> C.CDLL:
    """Load (once) and return the native library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950).  svi_hip has no CPU or PyTorch fallback.")
        l = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(l, name)          # AttributeError here == ABI drift; let it surface
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def last_error() -> str:
    return (lib().svi_last_error() or b"").decode("utf-8", "replace")


def check(status: int, what: str = "") -> None:
    if status != SVI_OK:
        raise RuntimeError(f"svi_hip{': ' + what if what else ''} failed (status {status}): {last_error()}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


_switch_epoch = 0


def switch_epoch() -> int:
    """How many times set_switch() has re-read the library's switches in this process.  Captured step graphs carry it in their key: a graph recorded
    under other switches (SVI_ATTN_QK8 changes the arithmetic, the rest the kernels) is never replayed."""
    return _switch_epoch


def set_switch(name: str, value=None) -> None:
    """A/B tooling: set (or, with None, remove) one of the library's environment switches and make the library re-read them."""
    global _switch_epoch
    _switch_epoch += 1
    if value is None:
        os.environ.pop(name, None)
    else:
        os.environ[name] = str(value)
    check(lib().svi_switches_reload(), "svi_switches_reload")


def release_stream_buffers(stream=None) -> None:
    """Free the library's per-stream device buffers of `stream` (a torch.cuda.Stream that is being retired); None: of every stream of the
    current device.  Captured graphs of that stream must not be replayed afterwards (DenoiseLoop checks svi_dit_generation and re-captures)."""
    check(lib().svi_stream_buffers_release(_vp(stream.cuda_stream) if stream is not None else None, 1 if stream is None else 0), "svi_stream_buffers_release")


def prof_enable(on: bool) -> None:
    check(lib().svi_prof_enable(1 if on else 0), "svi_prof_enable")


def gemm_plan(M: int, N: int, K: int, epilogue: int = 0, skinny: bool = False, compute_units: int = 256) -> int:
    """The kernel svi_gemm_bf16 would take on a part with `compute_units` CUs: 0 = weight-streaming skinny kernel, 128 = 128^2 tile, 192 = 256 x 192 tile,
    259 / 260 = the 256^2 tile with four / two phases per K tile (see include/svi_hip.h; ids 256-258 of rounds 1-3 are retired and refused with a warning)."""
    out = _i32(0)
    check(lib().svi_gemm_plan(M, N, K, epilogue, 1 if skinny else 0, compute_units, C.byref(out)), "svi_gemm_plan")
    return out.value


def attention_plan(s_q: int, s_kv: int, heads: int, compute_units: int = 256) -> dict:
    out = (_i32 * 4)()
    check(lib().svi_attention_plan(s_q, s_kv, heads, compute_units, out), "svi_attention_plan")
    return {"kernel": out[0], "whole": out[1], "pieces": out[2], "workgroups": out[3]}


def prof_select(tags=None) -> None:
    """Record only these tags (an iterable of names as prof_summary reports them); None = all."""
    arg = None if not tags else ",".join(tags).encode()
    check(lib().svi_prof_select(arg), "svi_prof_select")


def prof_summary() -> dict:
    """{tag: {"count": launches, "ms": total milliseconds}} for the launches since prof_enable(True)."""
    import json
    buf = C.create_string_buffer(8192)
    check(lib().svi_prof_summary(buf, len(buf)), "svi_prof_summary")
    return json.loads(buf.value.decode())
