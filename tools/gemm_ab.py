"""A/B the two GEMM kernels (128^2 register-staged vs 256^2 LDS-DMA) on the C2 shapes, interleaved in one process.

    python tools/gemm_ab.py [rounds]
For every shape: bit-compare the two kernels' outputs, then time them alternately (median / min over rounds).
The kernel is chosen per launch by the SVI_GEMM_KERNEL environment variable ("128" / "256"), read in svi_launch_gemm.
"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd"))
import torch  # noqa: E402

from svi_hip import _lib as L  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
Ltok, D, F = 32760, 1536, 8960
lib = L.lib()
st = L.current_stream()


def rnd(*shape, scale=1.0):
    return (torch.randn(shape, generator=g, device=dev) * scale).to(torch.bfloat16)


KINDS = tuple(os.environ.get("GEMM_AB_KINDS", "128,259,260").split(","))
SHAPES = {
    # name: (M, N, K, epilogue, bias_along_m)
    "qkv   [L,D]x[D,D]": (Ltok, D, D, L.EPI_BIAS, 0),
    "attn_o gate+res": (Ltok, D, D, L.EPI_BIAS_GATE_RES, 0),
    "v^T   [D,D]x[D,L]": (D, Ltok, D, L.EPI_BIAS, 1),
    "ffn1  gelu": (Ltok, F, D, L.EPI_BIAS_GELU_TANH, 0),
    "ffn2  gate+res": (Ltok, D, F, L.EPI_BIAS_GATE_RES, 0),
    "ragged 1000x520x192": (1000, 520, 192, L.EPI_BIAS_GELU_TANH, 0),
}
if len(sys.argv) > 2 and sys.argv[2] == "calib":       # calibration shapes against cdna_hip_programming.md's 256^2 numbers
    SHAPES = {"8192^3": (8192, 8192, 8192, L.EPI_BIAS, 0), "4096^3": (4096, 4096, 4096, L.EPI_BIAS, 0),
              "ffn2 M=8192": (8192, D, F, L.EPI_BIAS, 0), "ffn2 full, bias only": (Ltok, D, F, L.EPI_BIAS, 0),
              "N=6144 K=8960": (Ltok, 6144, F, L.EPI_BIAS, 0)}
if len(sys.argv) > 2 and sys.argv[2] == "shards":      # a sequence-parallel rank's token shard (L / P rows): where do the 256^2 tiles fill the chip poorly?
    SHAPES = {}
    for P in (2, 3, 4, 6):
        M = Ltok // P
        SHAPES[f"P={P} qkv"] = (M, D, D, L.EPI_BIAS, 0)
        SHAPES[f"P={P} ffn1"] = (M, F, D, L.EPI_BIAS_GELU_TANH, 0)
        SHAPES[f"P={P} ffn2"] = (M, D, F, L.EPI_BIAS_GATE_RES, 0)
for name, (M, N, K, epi, bam) in SHAPES.items():
    x = rnd(M, K); w = rnd(N, K, scale=K ** -0.5); b = rnd(M if bam else N)
    ldc = (N + 7) // 8 * 8
    gate = torch.randn(N, generator=g, device=dev)
    res = rnd(M, ldc)
    outs = {}

    def run(kind, out):
        L.set_switch("SVI_GEMM_KERNEL", kind)
        L.check(lib.svi_gemm_bf16(x.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), ldc, M, N, K, b.data_ptr(), bam, epi,
                                  gate.data_ptr() if epi == L.EPI_BIAS_GATE_RES else None,
                                  res.data_ptr() if epi == L.EPI_BIAS_GATE_RES else None, ldc, st))

    for kind in KINDS:
        outs[kind] = torch.zeros((M, ldc), dtype=torch.bfloat16, device=dev)
        run(kind, outs[kind])
    torch.cuda.synchronize()
    a = outs[KINDS[0]][:, :N].float()
    nbad = max(int((a != outs[k][:, :N].float()).sum()) for k in KINDS[1:])
    maxd = max(float((a - outs[k][:, :N].float()).abs().max()) for k in KINDS[1:])
    times = {k: [] for k in KINDS}
    for _ in range(rounds):
        for kind in KINDS:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run(kind, outs[kind])
            e1.record(); torch.cuda.synchronize()
            times[kind].append(e0.elapsed_time(e1) / 3)
    fl = 2.0 * M * N * K
    msg = f"{name:22s} M={M} N={N} K={K}  mismatching={nbad} maxdiff={maxd:.3g}"
    for kind in KINDS:
        med, mn = statistics.median(times[kind]), min(times[kind])
        msg += f" | k{kind}: med {med*1e3:.0f} us {fl/med/1e9:.0f} TF, best {fl/mn/1e9:.0f} TF"
    print(msg, flush=True)
L.set_switch("SVI_GEMM_KERNEL")
