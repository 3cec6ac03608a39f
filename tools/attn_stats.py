"""The self-attention kernel of the headline step on operands that are NOT the benign random-init case.

    python tools/attn_stats.py [--json PATH] [--quick]

The kernel measured is the instance the DiT launches 59 times per step — svi_attention_vt_fwd(q pre-scaled by softmax_scale*log2e, K token-major, V^T) at
[L = 32760, 12 heads x 128] = flash_fwd2_kernel<0, 0, false, 1> (optimistic pass) + <.., 2> (flagged second pass) — timed with HIP events on the launch stream.

Part A, "power or issue": the SAME binary on zero / small-magnitude / unit-Gaussian operands.  The instruction stream is identical (no data-dependent
branch in the optimistic pass), so any difference in ms per launch is the part's clock under its power limit (MI355X_MICROARCH.md "DVFS give-back").  The
shader clock is sampled from the driver's sysfs / rocm-smi while the kernel loops, when the box lets us.

Part B, "peaky logits": q and k rows are RMS-normalised per head as norm_q / norm_k leave them, times a gain g (the learned RMSNorm weight of a trained
checkpoint; random init has g = 1), plus a shared cluster direction of weight beta (tokens of one cluster — a contiguous run of `cluster` tokens, i.e. a
space-time neighbourhood in the (f h w) order — point the same way: attention to the own neighbourhood, which is what trained video DiTs show).  For every
(g, beta): on sampled query rows, in fp32 torch, the row entropy of the softmax and the OUTGROWTH = (row maximum - maximum over the first 64 keys) in log2
units — the optimistic pass fixes its reference after the first key tile and flags a workgroup whose row sums leave 2^64 — then the kernel: flagged
workgroups, ms per launch (both passes), TFLOP/s.
"""
import ctypes as C
import glob
import json
import math
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd"))
import torch  # noqa: E402

from svi_hip import _lib as L  # noqa: E402

args = sys.argv[1:]
json_path = None
if "--json" in args:
    i = args.index("--json")
    json_path = args[i + 1]
    del args[i:i + 2]
quick = "--quick" in args
dev = torch.device("cuda")
Ltok, H, DH = 32760, 12, 128
D = H * DH
FLOP = 4.0 * Ltok * Ltok * D
lib = L.lib()
C2E = (DH ** -0.5) * 1.4426950408889634


def sclk_mhz():
    """current shader clock from sysfs (pp_dpm_sclk marks the active level with '*'), or None"""
    for p in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for line in open(p).read().splitlines():
                if line.strip().endswith("*"):
                    return float(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
        except Exception:  # noqa: BLE001
            pass
    return None


def smi_sample():
    """(sclk MHz, socket power W) through rocm-smi, or (None, None)"""
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        j = json.loads(out)
        card = next(iter(j.values()))
        clk = pw = None
        for k, v in card.items():
            kl = k.lower()
            if "sclk" in kl and "(" in str(v):
                clk = float(str(v).split("(")[1].split("M")[0])
            if "power" in kl and "socket" in kl or "average graphics package power" in kl:
                try:
                    pw = float(v)
                except Exception:  # noqa: BLE001
                    pass
        return clk, pw
    except Exception:  # noqa: BLE001
        return None, None


def run_attention(q, k, vt, out):
    L.check(lib.svi_attention_vt_fwd(L.ptr(q), D, L.ptr(k), D, L.ptr(vt), Ltok, L.ptr(out), D, Ltok, Ltok, H, 1, L.current_stream()), "svi_attention_vt_fwd")


def last_flagged():
    a, b = C.c_int32(), C.c_int32()
    L.check(lib.svi_attention_last_flagged(L.current_stream(), C.byref(a), C.byref(b)), "svi_attention_last_flagged")
    return a.value, b.value


def timed(q, k, vt, out, launches, sample_clock=False):
    run_attention(q, k, vt, out)
    torch.cuda.synchronize()
    clocks, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            c = sclk_mhz()
            if c is not None:
                clocks.append(c)
            time.sleep(0.01)
    th = None
    if sample_clock and sclk_mhz() is not None:
        th = threading.Thread(target=poll, daemon=True)
        th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        run_attention(q, k, vt, out)
    e1.record()
    smi = smi_sample() if sample_clock else (None, None)          # taken while the launches are still queued / running
    torch.cuda.synchronize()
    stop.set()
    if th is not None:
        th.join()
    ms = e0.elapsed_time(e1) / launches
    return ms, (sum(clocks) / len(clocks) if clocks else None), smi


rows = []
g = torch.Generator(device=dev).manual_seed(0)
out = torch.empty((Ltok, D), dtype=torch.bfloat16, device=dev)
NL = 6 if quick else 16

# ---- part A: the same instruction stream on operands of different toggle activity ------------------------------------------------------------
for name, scale in (("zeros", 0.0), ("small (x 1e-3)", 1e-3), ("unit Gaussian (the benign case of the bench)", 1.0)):
    q = (torch.randn((Ltok, D), generator=g, device=dev) * scale * C2E).to(torch.bfloat16)
    k = (torch.randn((Ltok, D), generator=g, device=dev) * scale).to(torch.bfloat16)
    vt = (torch.randn((D, Ltok), generator=g, device=dev) * scale).to(torch.bfloat16)
    ms, clk, smi = timed(q, k, vt, out, NL, sample_clock=True)
    flagged, nwg = last_flagged()
    row = {"part": "A", "operands": name, "ms_per_launch": round(ms, 4), "tflops": round(FLOP / ms / 1e9, 1), "frac_of_2500": round(FLOP / ms / 1e9 / 2500.0, 4),
           "flagged": flagged, "workgroups": nwg, "sclk_mhz_sysfs_mean": clk, "rocm_smi_sclk_mhz": smi[0], "rocm_smi_power_w": smi[1]}
    rows.append(row)
    print(json.dumps(row), flush=True)
    del q, k, vt

if "--a-only" in args:
    sys.exit(0)
# ---- part B: peaky logits ----------------------------------------------------------------------------------------------------------------------
cluster = 1560                                   # one latent frame of the C2 grid (30 x 52 tokens): a query's own frame is its neighbourhood
ncl = Ltok // cluster
vt = torch.randn((D, Ltok), generator=g, device=dev).to(torch.bfloat16)
sample_rows = torch.randint(0, Ltok, (384,), generator=g, device=dev)


def make(gain, beta):
    z = torch.randn((2, Ltok, H, DH), generator=g, device=dev)
    c = torch.randn((ncl + 1, H, DH), generator=g, device=dev)
    cid = (torch.arange(Ltok, device=dev) // cluster).clamp(max=ncl)
    z = z + beta * c[cid][None]
    z = z / z.pow(2).mean(dim=-1, keepdim=True).sqrt()          # unit RMS per head row, as RMSNorm (full width) leaves a row on average
    z = z * gain
    q = (z[0].reshape(Ltok, D) * C2E).to(torch.bfloat16)
    k = z[1].reshape(Ltok, D).to(torch.bfloat16)
    return q.contiguous(), k.contiguous()


def row_stats(q, k):
    """fp32 on the sampled rows of head 0 and head H-1: entropy (nats) and outgrowth over the first 64 keys (log2 units)"""
    ent, outg = [], []
    for hd in (0, H - 1):
        qs = q[sample_rows, hd * DH:(hd + 1) * DH].float()
        ks = k[:, hd * DH:(hd + 1) * DH].float()
        s = qs @ ks.t()                                     # log2 units (q carries scale * log2e)
        mx = s.max(dim=1).values
        outg.append(mx - s[:, :64].max(dim=1).values)
        p = torch.softmax(s * math.log(2.0), dim=1)
        ent.append(-(p * (p.clamp_min(1e-30)).log()).sum(dim=1))
    ent, outg = torch.cat(ent), torch.cat(outg)
    return float(ent.mean()), float(ent.min()), float(outg.max()), float(outg.mean())


grid = [(1.0, 0.0), (1.5, 0.0), (2.0, 0.0), (3.0, 0.0), (1.0, 0.5), (1.5, 0.5), (2.0, 0.5), (1.0, 1.0), (1.5, 1.0), (2.0, 1.0), (2.5, 1.0), (3.0, 1.0), (4.0, 1.0)]
if quick:
    grid = grid[::3]
for gain, beta in grid:
    q, k = make(gain, beta)
    ent_mean, ent_min, outg_max, outg_mean = row_stats(q, k)
    ms, _, _ = timed(q, k, vt, out, NL)
    flagged, nwg = last_flagged()
    finite = bool(torch.isfinite(out.float()).all().item())
    row = {"part": "B", "gain": gain, "beta": beta, "row_entropy_nats_mean": round(ent_mean, 3), "row_entropy_nats_min": round(ent_min, 3),
           "uniform_entropy_nats": round(math.log(Ltok), 3), "outgrowth_log2_max": round(outg_max, 2), "outgrowth_log2_mean": round(outg_mean, 2),
           "flagged": flagged, "workgroups": nwg, "flagged_fraction": round(flagged / max(nwg, 1), 4), "ms_per_launch_both_passes": round(ms, 4),
           "tflops": round(FLOP / ms / 1e9, 1), "finite": finite}
    rows.append(row)
    print(json.dumps(row), flush=True)
    del q, k

if json_path:
    os.makedirs(os.path.dirname(os.path.abspath(json_path)), exist_ok=True)
    with open(json_path, "w") as f:
        json.dump({"device": torch.cuda.get_device_name(0), "L": Ltok, "heads": H, "flop_per_launch": FLOP, "rows": rows}, f, indent=1)
