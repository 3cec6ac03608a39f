"""Run ONE hot kernel at the C2 problem size in isolation (for rocprofv3 --pmc / --kernel-trace passes).

    python tools/kernel_probe.py attn [iters]      self-attention flash kernel, L=32760, 12 heads
    python tools/kernel_probe.py gemm_ffn1|gemm_ffn2|gemm_qkv [iters]
    python tools/kernel_probe.py ln|rms [iters]
    python tools/kernel_probe.py cross [iters]     the fused cross-attention (q RMS-normalised as it is read; 65 of 512 keys, resident in LDS) at L = 32760
    python tools/kernel_probe.py vae [iters]       VAE decode of 3 latent frames at 480x832 (every decoder layer at its C2 spatial size)
Prints achieved TFLOP/s (or TB/s) from torch.cuda.Event timing on the launch stream.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd"))
import torch  # noqa: E402

import svi_hip  # noqa: E402
from svi_hip import _lib as L  # noqa: E402

what = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
Ltok, D, F, H = 32760, 1536, 8960, 12


def rnd(*shape, scale=1.0):
    return (torch.randn(shape, generator=g, device=dev) * scale).to(torch.bfloat16)


def timeit(fn, flops=None, bytes_=None):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    msg = f"{what}: {ms:.4f} ms/launch"
    if flops:
        msg += f"  {flops / ms / 1e9:.1f} TFLOP/s"
    if bytes_:
        msg += f"  {bytes_ / ms / 1e9:.3f} TB/s"
    print(msg)


lib = L.lib()
st = L.current_stream()
if what == "attn":
    qk = rnd(Ltok, 2 * D)
    vt = rnd(D, Ltok)
    out = torch.empty((Ltok, D), dtype=torch.bfloat16, device=dev)
    # the DiT-internal launch (q|k interleaved buffer, V^T) goes through svi_dit_block_forward; for the probe use the
    # public seam, which adds one transpose (reported separately by the trace)
    q = qk[:, :D].contiguous(); k = qk[:, D:].contiguous(); v = vt.t().contiguous()
    timeit(lambda: L.check(lib.svi_attention_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), 1, Ltok, Ltok, H, 128, st)),
           flops=4.0 * Ltok * Ltok * D)
elif what.startswith("gemm"):
    M, N, K, epi = {"gemm_ffn1": (Ltok, F, D, L.EPI_BIAS_GELU_TANH), "gemm_ffn2": (Ltok, D, F, L.EPI_BIAS_GATE_RES),
                    "gemm_qkv": (Ltok, D, D, L.EPI_BIAS)}[what]
    x = rnd(M, K); w = rnd(N, K, scale=K ** -0.5); b = rnd(N)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    gate = torch.randn(N, generator=g, device=dev)
    res = rnd(M, N)
    timeit(lambda: L.check(lib.svi_gemm_bf16(x.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, b.data_ptr(), 0, epi,
                                             gate.data_ptr() if epi == L.EPI_BIAS_GATE_RES else None,
                                             res.data_ptr() if epi == L.EPI_BIAS_GATE_RES else None, N, st)), flops=2.0 * M * N * K)
elif what == "ln":
    x = rnd(Ltok, D); sh = rnd(D); sc = rnd(D)
    timeit(lambda: svi_hip.layernorm_modulate(x, 1e-6, shift=sh, scale=sc), bytes_=2.0 * Ltok * D * 2)
elif what == "rms":
    x = rnd(Ltok, D); wt = rnd(D)
    timeit(lambda: svi_hip.rmsnorm_rope_(x, wt, 1e-6, grid=(21, 30, 52), num_heads=12), bytes_=2.0 * Ltok * D * 2)
elif what == "cross":
    # the DiT block's cross-attention query path at the C2 size: q projection with the row statistics, then the fused attention (65 keys walked of 512)
    x = rnd(Ltok, D); w = rnd(D, D, scale=D ** -0.5); b = rnd(D); gain = rnd(D)
    k = rnd(512, D); vt = rnd(D, 512)
    k[64:] = k[64]; vt[:, 64:] = vt[:, 64:65]
    tail = torch.tensor([65, 448], dtype=torch.int32, device=dev)
    y, rs, _ = svi_hip.ops.linear_row_stats(x, w, b, eps=1e-6)
    timeit(lambda: svi_hip.ops.cross_attention(y, k, vt, H, s_kv=512, q_rs=rs, q_gain=gain, q_out_scale=0.12751743, key_tail=tail), bytes_=2.0 * Ltok * D * 2)
elif what == "vae":
    from svi_hip.vae import WanVideoVAE, device_vae_weights
    vae = WanVideoVAE.from_state_dict(device_vae_weights(0, dev))
    z = torch.randn((16, 3, 60, 104), generator=g, device=dev)
    timeit(lambda: vae.decode([z], device=dev)[0])
else:
    raise SystemExit("unknown probe")
