"""The whole-sequence VAE oracle against the reference's chunked/cached implementation (golden)."""
import numpy as np
import torch

import synth
from conftest import rel_l2
from oracle import wan_vae_oracle as wvo


def _sd():
    return {k: torch.from_numpy(v) for k, v in synth.vae_state_dict(500).items()}


def test_param_inventory():
    shapes = synth.vae_param_shapes()
    assert len(shapes) == 194
    assert sum(int(np.prod(s)) for s in shapes.values()) == 126892531
    n_conv3 = lambda pre: sum(1 for k, s in shapes.items() if k.startswith(pre) and k.endswith("weight") and len(s) == 5)
    assert n_conv3("model.encoder.") == 26 and n_conv3("model.decoder.") == 33


def test_decode_matches_reference(golden):
    g, sd = golden("vae.npz"), _sd()
    with torch.no_grad():
        for key, seed, shape in (("decode_3f", 501, (1, 16, 3, 4, 6)), ("decode_1f", 502, (1, 16, 1, 4, 6)),
                                 ("decode_2f_tinyhw", 505, (1, 16, 2, 2, 2))):
            out = wvo.vae_decode(sd, torch.from_numpy(synth.randn(seed, *shape)))[0].numpy()
            assert out.shape == g[key].shape
            assert rel_l2(out, g[key]) < 2e-5, key
            assert np.abs(out - g[key]).max() < 1e-4, key


def test_encode_matches_reference(golden):
    g, sd = golden("vae.npz"), _sd()
    with torch.no_grad():
        for key, seed, shape in (("encode_9f", 503, (3, 9, 32, 48)), ("encode_1f", 504, (3, 1, 32, 48))):
            vid = torch.from_numpy(np.tanh(synth.randn(seed, *shape)))[None]
            out = wvo.vae_encode(sd, vid)[0].numpy()
            assert out.shape == g[key].shape
            assert rel_l2(out, g[key]) < 2e-5, key
