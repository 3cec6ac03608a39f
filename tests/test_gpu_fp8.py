"""-m gpu: FP8 weight storage (BASELINE configs[4]).  The reference's FP8 mode keeps parameters as float8_e4m3fn and casts
them to bf16 in front of every use; svi_hip does the (exact) cast once at bind time.  Checked: the decode kernel on all 256 code points
against torch's own cast (bit-exact), and a forward on fp8-stored weights against the reference's forward on bf16(e4m3(W)) weights."""
import numpy as np
import pytest
import torch

import synth
from gpu_util import dev, errs, report
from test_oracle_dit import CASES, inputs

pytestmark = pytest.mark.gpu


def test_e4m3_decode_is_torchs_cast(golden):
    import svi_hip
    from svi_hip.ops import fp8_e4m3_to_bf16
    codes = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn)
    got = fp8_e4m3_to_bf16(codes.cuda()).cpu().view(torch.int16).numpy()
    want = golden("fp8_storage.npz")["e4m3_to_bf16_bits"]
    nan = np.isnan(codes.to(torch.float32).numpy())
    assert np.array_equal(got[~nan], want[~nan])
    assert np.isnan(torch.from_numpy(got[nan]).view(torch.bfloat16).float().numpy()).all()
    big = torch.randint(0, 256, (3, 1000, 77), dtype=torch.uint8).view(torch.float8_e4m3fn)
    a, b = fp8_e4m3_to_bf16(big.cuda()).cpu().float(), big.to(torch.bfloat16).float()
    assert torch.equal(torch.nan_to_num(a, nan=7.0), torch.nan_to_num(b, nan=7.0))


def test_forward_on_fp8_stored_weights_matches_reference(golden):
    import svi_hip
    c, grid, nt, nv, ts, seed = CASES["tiny_t2v"]
    sd = {k: torch.from_numpy(v).to(torch.float8_e4m3fn) for k, v in synth.dit_state_dict(seed, **c).items()}
    m8 = svi_hip.WanDiT.from_state_dict(sd, eps=1e-6, num_heads=synth.num_heads_of(c), **c)
    assert len(m8._fp8_sources) == len(sd)
    m16 = svi_hip.WanDiT.from_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()}, eps=1e-6, num_heads=synth.num_heads_of(c), **c)
    x, ctx, _ = inputs(c, grid, nt, nv, seed)
    a = m8.forward(dev(x), torch.tensor([ts]), dev(ctx))
    b = m16.forward(dev(x), torch.tensor([ts]), dev(ctx))
    assert torch.equal(a, b)                                   # the bind-time cast is torch's cast
    r = errs(a, golden("fp8_storage.npz")["out_bf16"])[0]
    r_plain = errs(a, golden("dit_tiny_t2v.npz")["out_bf16"])[0]
    report("fp8_storage", vs_reference_fp8_mode=r, vs_reference_bf16_mode=r_plain)
    assert r < 2e-2 and r_plain > 2 * r, (r, r_plain)          # and it is the fp8-mode result, not the bf16-mode one
