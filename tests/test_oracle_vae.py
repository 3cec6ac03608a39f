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


def test_tiled_paths_match_reference(golden):
    """tiled_decode / tiled_encode (vae:643-744) through the reference's public decode/encode(tiled=True): task list with clipped
    tiles, ramp masks, blend order, clamp after the blend (1.2 % of the golden pixels sit on the clamp)."""
    g, sd = golden("vae_tiled.npz"), _sd()
    with torch.no_grad():
        for name, zshape, size, stride, seed in synth.TILED_DECODE_CASES:
            out = wvo.tiled_decode(sd, torch.from_numpy(2.0 * synth.randn(seed, *zshape))[None], size, stride)[0].numpy()
            want = g["decode_" + name]
            assert out.shape == want.shape
            assert rel_l2(out, want) < 2e-5 and np.abs(out - want).max() < 2e-4, name
        for name, vshape, size, stride, seed in synth.TILED_ENCODE_CASES:
            vid = torch.from_numpy(np.tanh(synth.randn(seed, *vshape)))[None]
            out = wvo.tiled_encode(sd, vid, (size[0] * 8, size[1] * 8), (stride[0] * 8, stride[1] * 8))[0].numpy()
            want = g["encode_" + name]
            assert out.shape == want.shape
            assert rel_l2(out, want) < 2e-5, name


def test_tile_task_list_and_masks():
    """The default tiling of the pipelines at 81f 832x480 (tile (30,52), stride (15,26) on a 60x104 latent, svi_video.py:439-440):
    9 tiles; masks: ones at bound edges, (i+1)/border ramps elsewhere, the right ramp written last."""
    tasks = wvo.tile_tasks(60, 104, (30, 52), (15, 26))
    assert tasks == [(h, h + 30, w, w + 52) for h in (0, 15, 30) for w in (0, 26, 52)]
    m = wvo.ramp_1d(8, False, True, 4).numpy()
    assert np.array_equal(m, np.array([0.25, 0.5, 0.75, 1, 1, 1, 1, 1], np.float32))
    m = wvo.ramp_1d(6, False, False, 4).numpy()          # overlapping ramps: the right one wins
    assert np.array_equal(m, np.array([0.25, 0.5, 1.0, 0.75, 0.5, 0.25], np.float32))


def test_c2_spatial_size_matches_reference(golden):
    """The whole-sequence oracle at BASELINE config 2's spatial size (latent 60x104 <-> 480x832 px), against the reference's chunked /
    cached implementation (golden/vae_c2.npz): 2 latent frames decoded, 5 frames encoded."""
    g, sd = golden("vae_c2.npz"), _sd()
    k = synth.C2_VIDEO_STRIDE
    with torch.no_grad():
        out = wvo.vae_decode(sd, torch.from_numpy(synth.randn(511, 16, 2, 60, 104))[None])[0][:, :, ::k, ::k].numpy()
        assert out.shape == g["decode_sample"].shape
        assert rel_l2(out, g["decode_sample"]) < 2e-5 and np.abs(out - g["decode_sample"]).max() < 2e-4
        lat = wvo.vae_encode(sd, torch.from_numpy(np.tanh(synth.randn(512, 3, 5, 480, 832)))[None])[0].numpy()
        assert rel_l2(lat, g["encode"]) < 2e-5
