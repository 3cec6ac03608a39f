"""Host logic of SURVEY §8f N4: which LoRA tensors patch which parameter (against the reference's own get_name_dict, golden/lora_names.json)
and the safetensors reader (CPU round trip; the device path is the same call with device="cuda")."""
import json
import os

import torch

import synth
from conftest import GOLDEN


def test_lora_name_pairs_match_the_reference():
    from svi_hip import lora
    want = {k: tuple(v) for k, v in json.load(open(os.path.join(GOLDEN, "lora_names.json"))).items()}
    assert lora.name_pairs(synth.LORA_KEY_EXAMPLES) == want
    assert set(want) == {"blocks.0.self_attn.q.weight", "blocks.1.self_attn.o.weight", "blocks.29.cross_attn.k_img.weight", "blocks.3.ffn.0.weight"}


def test_safetensors_shards_round_trip(tmp_path):
    from safetensors.torch import save_file
    from svi_hip import checkpoint
    a = {"x.weight": torch.arange(12, dtype=torch.float32).reshape(3, 4), "y.bias": torch.ones(5, dtype=torch.bfloat16)}
    b = {"z.weight": torch.full((2, 2), 3.0), "y.bias": torch.zeros(5, dtype=torch.bfloat16)}          # a later shard overrides a key
    save_file(a, str(tmp_path / "a.safetensors"))
    save_file(b, str(tmp_path / "b.safetensors"))
    sd = checkpoint.load_safetensors([str(tmp_path / "a.safetensors"), str(tmp_path / "b.safetensors")], device="cpu")
    assert set(sd) == {"x.weight", "y.bias", "z.weight"} and torch.equal(sd["x.weight"], a["x.weight"]) and not sd["y.bias"].any()
    sd16 = checkpoint.load_safetensors(str(tmp_path / "a.safetensors"), device="cpu", torch_dtype=torch.bfloat16)
    assert all(t.dtype == torch.bfloat16 for t in sd16.values())
