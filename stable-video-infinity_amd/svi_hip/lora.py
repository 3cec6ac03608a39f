"""LoRA merge on the device (the arithmetic of models/lora.py:246-262).

The reference patches every targeted weight at load time:  W <- W + alpha * (up @ down), computed in the model's dtype (bf16):
`torch.mm` rounds the product to bf16, the scaling by alpha rounds again, the addition rounds again.  That is exactly the
GEMM's gate / residual epilogue — bf16(res + bf16(gate * bf16(acc))) — so a merge is one launch per matrix, in place, with the
reference's rounding points (the fp32 accumulation order inside the 128-long dot products is the only freedom left).  For the
14B model that is ~400 launches of a [5120, r] x [r, 5120] product instead of ~400 CPU matrix products.

Which tensors pair up (`lora_A/lora_B`, `lora_up/lora_down`, prefixes) is the reference loader's business (get_name_dict); this
module takes the pairs.  Weights borrowed by a WanDiT stay valid (the update is in place); the context cache, whose cross-attention
K / V were projected with the old weights, is dropped: pass `dit=` to merge_state_dict_ (it calls `dit.rebind()`), and a WanDiT also
notices in-place writes to its bound parameters by itself (version counters, WanDiT.weights_changed) before the next forward.
"""
from __future__ import annotations

from typing import Dict, Iterable, Tuple

import torch

from . import _lib as L


def merge_lora_(weight: torch.Tensor, up: torch.Tensor, down: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """weight [out, in] bf16 on the GPU, updated in place:  weight += alpha * up[out, r] @ down[r, in]."""
    if not (weight.is_cuda and weight.dtype == torch.bfloat16 and weight.is_contiguous() and weight.dim() == 2):
        raise ValueError("weight must be a contiguous 2-D CUDA bf16 tensor")
    out_f, in_f = weight.shape
    up = up.reshape(up.shape[0], -1).to(device=weight.device, dtype=torch.bfloat16).contiguous()          # conv-style [o, r, 1, 1] too
    down_t = down.reshape(down.shape[0], -1).to(device=weight.device, dtype=torch.bfloat16).t().contiguous()   # [in, r]
    r = up.shape[1]
    if up.shape[0] != out_f or down_t.shape != (in_f, r):
        raise ValueError(f"LoRA pair {tuple(up.shape)} x {tuple(down.shape)} does not match weight {tuple(weight.shape)}")
    if r % 8 or in_f % 8:
        raise ValueError("rank and in_features must be multiples of 8")
    gate = torch.full((in_f,), float(alpha), dtype=torch.float32, device=weight.device)
    L.check(L.lib().svi_gemm_bf16(L.ptr(up), r, L.ptr(down_t), r, L.ptr(weight), in_f, out_f, in_f, r, None, 0, L.EPI_BIAS_GATE_RES,
                                  L.ptr(gate), L.ptr(weight), in_f, L.current_stream()), "lora merge")
    weight[:0].zero_()          # the kernel wrote through a raw pointer: bump the tensor's version counter (no launch for 0 elements)
    return weight


def merge_state_dict_(state_dict: Dict[str, torch.Tensor], pairs: Dict[str, Tuple[torch.Tensor, torch.Tensor]], alpha: float = 1.0,
                      dit=None) -> int:
    """Patch state_dict[name] in place for every name -> (up, down) pair; returns the number of tensors updated
    (the reference prints it, lora.py:263).  `dit`: the WanDiT that borrows these tensors — re-bound afterwards."""
    for name, (up, down) in pairs.items():
        merge_lora_(state_dict[name], up, down, alpha)
    if dit is not None:
        dit.rebind()
    return len(pairs)


def name_pairs(lora_keys: Iterable[str]) -> Dict[str, Tuple[str, str]]:
    """Which LoRA tensors patch which parameter — GeneralLoRAFromPeft.get_name_dict (models/lora.py:204-217): for every key holding
    ".lora_B." the target parameter name is the key without the "lora_B" component, without the adapter name that may follow it,
    and without a leading "diffusion_model"; the pair is (up = the lora_B key, down = the same key with lora_A)."""
    out: Dict[str, Tuple[str, str]] = {}
    for key in lora_keys:
        if ".lora_B." not in key:
            continue
        parts = key.split(".")
        i = parts.index("lora_B")
        if len(parts) > i + 2:
            parts.pop(i + 1)
        parts.pop(i)
        if parts[0] == "diffusion_model":
            parts.pop(0)
        out[".".join(parts)] = (key, key.replace(".lora_B.", ".lora_A."))
    return out


def load_lora_(dit, lora_state_dict: Dict[str, torch.Tensor], alpha: float = 1.0) -> int:
    """ModelManager.load_lora_v2 -> GeneralLoRAFromPeft.load (models/lora.py:246-267) for a WanDiT: every matched parameter is patched
    in place on the device (one GEMM launch each), the handle re-bound.  Returns the number of tensors updated (the reference prints it).
    Raises if a LoRA target is not a parameter of the model (the reference's `match` would have refused the file)."""
    pairs = name_pairs(lora_state_dict.keys())
    missing = [n for n in pairs if n not in dit._params]
    if missing:
        raise KeyError(f"LoRA targets that are not parameters of this model: {missing[:3]}{' ...' if len(missing) > 3 else ''}")
    if dit._fp8_sources:
        raise NotImplementedError("LoRA merge into fp8-stored parameters (the reference merges in fp32 and re-quantises) is not served")
    return merge_state_dict_(dit._params, {n: (lora_state_dict[u], lora_state_dict[d]) for n, (u, d) in pairs.items()}, alpha, dit=dit)
