"""Vendor yardstick: the six hot shapes of the C2 step on the vendor libraries, beside this repo's kernels, in ONE process on ONE box.

    python tools/yardstick.py [rounds] [--json PATH]

tools only — never product: the package does not link, load or call hipBLASLt / rocBLAS / any SDPA backend; this script asks PyTorch-ROCm for them so
that "0.4-0.6 of peak" has something to stand beside (VERDICT r5 "missing 2").  Shapes (wan_video_dit.py:227-229,242,334-335,116-147):
    q|k     [32760, 1536] x [3072, 1536]^T      v      [32760, 1536] x [1536, 1536]^T      attn-out / cross q,o   the same 1536^2 weight, gate + residual
    ffn1    [32760, 1536] x [8960, 1536]^T + GELU(tanh)        ffn2   [32760, 8960] x [1536, 8960]^T, gate + residual
    self-attention  [1, 12, 32760, 128]
For each GEMM: (a) torch.matmul = the bare vendor GEMM, nothing fused; (b) the module as PyTorch-ROCm would run it (F.linear with bias, then the
activation / gate / residual as separate elementwise kernels) = the "fallback" the north star says not to be; (c) this repo's launch with everything
fused.  Timed interleaved, median of `rounds` groups of 3 launches, HIP events on the launch stream.  Operands: uniform-free Gaussian bf16 (weights
scaled K^-0.5) — the guide's rule: bench on random data.
"""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import svi_hip  # noqa: E402
from svi_hip import _lib as L  # noqa: E402

args = [a for a in sys.argv[1:]]
json_path = None
if "--json" in args:
    i = args.index("--json")
    json_path = args[i + 1]
    del args[i:i + 2]
rounds = int(args[0]) if args else 7
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
Ltok, D, Fd, H = 32760, 1536, 8960, 12
lib = L.lib()
st = L.current_stream()


def rnd(*shape, scale=1.0):
    return (torch.randn(shape, generator=g, device=dev) * scale).to(torch.bfloat16)


def timed(fns, rounds):
    """fns: {name: callable}; interleaved groups of 3 launches each; returns {name: (median ms, min ms)}"""
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    ts = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                f()
            e1.record()
            torch.cuda.synchronize()
            ts[k].append(e0.elapsed_time(e1) / 3)
    return {k: (statistics.median(v), min(v)) for k, v in ts.items()}


rows = []
SHAPES = [
    # name, M, N, K, epilogue
    ("qk", Ltok, 2 * D, D, L.EPI_BIAS),
    ("v", Ltok, D, D, L.EPI_BIAS),
    ("attn_out", Ltok, D, D, L.EPI_BIAS_GATE_RES),
    ("ffn1", Ltok, Fd, D, L.EPI_BIAS_GELU_TANH),
    ("ffn2", Ltok, D, Fd, L.EPI_BIAS_GATE_RES),
]
for name, M, N, K, epi in SHAPES:
    x = rnd(M, K)
    w = rnd(N, K, scale=K ** -0.5)
    b = rnd(N)
    gate = torch.randn(N, generator=g, device=dev)
    res = rnd(M, N)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    out_v = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    wt = w.t()
    gate_b = gate.to(torch.bfloat16)

    def ours():
        L.check(lib.svi_gemm_bf16(x.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, b.data_ptr(), 0, epi,
                                  gate.data_ptr() if epi == L.EPI_BIAS_GATE_RES else None,
                                  res.data_ptr() if epi == L.EPI_BIAS_GATE_RES else None, N, st))

    def vendor_bare():
        torch.matmul(x, wt, out=out_v)

    def vendor_module():
        y = F.linear(x, w, b)
        if epi == L.EPI_BIAS_GELU_TANH:
            y = F.gelu(y, approximate="tanh")
        elif epi == L.EPI_BIAS_GATE_RES:
            y = res + gate_b * y
        return y

    # value check of OUR launch against the vendor module (same bf16 operands; fp32 accumulate on both sides): rel-L2
    ours()
    ref = vendor_module().float()
    rel = float((out.float() - ref).norm() / ref.norm())
    t = timed({"ours": ours, "vendor_bare": vendor_bare, "vendor_module": vendor_module}, rounds)
    fl = 2.0 * M * N * K
    row = {"shape": name, "M": M, "N": N, "K": K, "flop": fl, "rel_l2_vs_vendor_module": rel}
    for k, (med, mn) in t.items():
        row[k + "_ms"] = round(med, 4)
        row[k + "_tf"] = round(fl / med / 1e9, 1)
        row[k + "_best_tf"] = round(fl / mn / 1e9, 1)
    rows.append(row)
    print(json.dumps(row), flush=True)
    del x, w, b, res, out, out_v

# self-attention [1, 12, 32760, 128]: this repo's seam takes [b, s, (n d)]; SDPA takes [b, n, s, d]
q, k, v = rnd(1, Ltok, H * 128), rnd(1, Ltok, H * 128), rnd(1, Ltok, H * 128)
qh = q.view(1, Ltok, H, 128).transpose(1, 2).contiguous()
kh = k.view(1, Ltok, H, 128).transpose(1, 2).contiguous()
vh = v.view(1, Ltok, H, 128).transpose(1, 2).contiguous()
fl = 4.0 * Ltok * Ltok * H * 128
fns = {"ours": lambda: svi_hip.flash_attention(q, k, v, H)}
row = {"shape": "self_attention", "b": 1, "heads": H, "L": Ltok, "d": 128, "flop": fl}
backends = {}
try:
    from torch.nn.attention import SDPBackend, sdpa_kernel
    for nm, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION)):
        def mk(be):
            def f():
                with sdpa_kernel(be):
                    return F.scaled_dot_product_attention(qh, kh, vh)
            return f
        try:
            o = mk(be)()
            torch.cuda.synchronize()
            backends["vendor_sdpa_" + nm] = mk(be)
            got = svi_hip.flash_attention(q, k, v, H).view(1, Ltok, H, 128).transpose(1, 2).float()
            row["rel_l2_vs_sdpa_" + nm] = float((got - o.float()).norm() / o.float().norm())
        except Exception as e:  # noqa: BLE001
            row["vendor_sdpa_" + nm + "_error"] = str(e).splitlines()[0][:200]
except Exception as e:  # noqa: BLE001
    row["sdpa_error"] = str(e)[:200]
if not backends:
    try:
        F.scaled_dot_product_attention(qh, kh, vh)
        backends["vendor_sdpa_default"] = lambda: F.scaled_dot_product_attention(qh, kh, vh)
    except Exception as e:  # noqa: BLE001
        row["vendor_sdpa_default_error"] = str(e).splitlines()[0][:200]
fns.update(backends)
t = timed(fns, max(3, rounds // 2))
for kk, (med, mn) in t.items():
    row[kk + "_ms"] = round(med, 4)
    row[kk + "_tf"] = round(fl / med / 1e9, 1)
    row[kk + "_best_tf"] = round(fl / mn / 1e9, 1)
rows.append(row)
print(json.dumps(row), flush=True)

summary = {"device": torch.cuda.get_device_name(0), "torch": torch.__version__, "hip": torch.version.hip, "rounds": rounds, "rows": rows}
if json_path:
    os.makedirs(os.path.dirname(os.path.abspath(json_path)), exist_ok=True)
    with open(json_path, "w") as f:
        json.dump(summary, f, indent=1)
print("| shape | ours ms (TF) | vendor bare ms (TF) | vendor module ms (TF) | ours / vendor bare |")
print("|---|---|---|---|---|")
for r in rows:
    if "ours_ms" not in r:
        continue
    vb = r.get("vendor_bare_ms") or r.get("vendor_sdpa_flash_ms") or r.get("vendor_sdpa_efficient_ms") or r.get("vendor_sdpa_default_ms")
    vbtf = r.get("vendor_bare_tf") or r.get("vendor_sdpa_flash_tf") or r.get("vendor_sdpa_efficient_tf") or r.get("vendor_sdpa_default_tf")
    vm = r.get("vendor_module_ms")
    print(f"| {r['shape']} | {r['ours_ms']:.3f} ({r['ours_tf']:.0f}) | {vb if vb else '-'} ({vbtf if vbtf else '-'}) | "
          f"{vm if vm else '-'} ({r.get('vendor_module_tf', '-')}) | {(vb / r['ours_ms']) if vb else float('nan'):.2f}x |")
