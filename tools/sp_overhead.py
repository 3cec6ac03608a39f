"""What the sequence-parallel schedule costs besides the transport, on ONE GPU: P shards run back to back with the exchange
simulated by device copies (svi_hip.sequence_parallel.forward_local) vs the plain forward, Wan2.1-1.3B widths, C2 geometry.
With P ranks on P GPUs the shard work runs concurrently, so  (local time / P)  is the per-rank compute + packing time that the
all-to-all transport is added to.   python tools/sp_overhead.py [layers] [tags] [pair]
`pair`: the CFG PAIR per step instead — the single-rank stacked pair (WanDiT.forward_cfg_pair) against (a) two plain shard forwards per rank
(forward_local twice: what a sequence-parallel rank ran before round 5) and (b) the stacked pair on every shard (forward_local_pair)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
import svi_hip, synth
from svi_hip import sequence_parallel as sp
from bench import device_weights
layers = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 6
dev = torch.device("cuda")
cfg = dict(synth.WAN_1_3B); cfg["num_layers"] = layers
sd = device_weights(cfg, 0, dev)
def handle():
    m = svi_hip.WanDiT(eps=1e-6, num_heads=12, **cfg); m.bind(sd); return m
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn((1, 16, 21, 60, 104), generator=g, device=dev).to(torch.bfloat16)
ctx = torch.randn((1, 512, 4096), generator=g, device=dev).to(torch.bfloat16)
t = torch.tensor([500.0])
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
plain = handle()
for m in (plain,): m.context_cache(True)
base = timeit(lambda: plain.forward(x, t, ctx))
print(f"plain forward, {layers} blocks: {base:.1f} ms")
from svi_hip import _lib
def tags(fn):
    """per-tag kernel time of one call (svi_prof_*), ms"""
    fn(); torch.cuda.synchronize()
    _lib.prof_enable(True); fn(); torch.cuda.synchronize()
    out = {k: v["ms"] for k, v in _lib.prof_summary().items() if v["ms"] > 0}
    _lib.prof_enable(False)
    return out
base_tags = tags(lambda: plain.forward(x, t, ctx))
if "pair" in sys.argv[1:]:
    ctx2 = torch.randn((1, 512, 4096), generator=g, device=dev).to(torch.bfloat16)
    ctx[:, 64:] = 0; ctx2[:, 32:] = 0
    pair_base = timeit(lambda: plain.forward_cfg_pair(x, t, ctx, ctx2))
    pair_tags = tags(lambda: plain.forward_cfg_pair(x, t, ctx, ctx2))
    print(f"single rank, stacked CFG pair, {layers} blocks: {pair_base:.1f} ms per step (two plain forwards: {2 * base:.1f} ms)")
    for P in (2, 4, 6):
        hs = [handle() for _ in range(P)]
        for m in hs: m.context_cache(True)
        G1 = sp.head_groups(12 // P, 32760, 1)
        def two(): sp.forward_local(hs, x, t, ctx, groups=G1); sp.forward_local(hs, x, t, ctx2, groups=G1)
        ms2 = timeit(two)
        for G in sorted({1, sp.head_groups(12 // P, 32760, 2)}):
            msp = timeit(lambda: sp.forward_local_pair(hs, x, t, ctx, ctx2, groups=G))
            def copies(nb):
                bufs = [m._sp_buffers[k] for m in hs for k in m._sp_buffers if k[3] == (G if nb == 2 else G1) and k[5] == nb]
                for _ in range(layers):
                    for j, bj in enumerate(bufs):
                        for i, bi in enumerate(bufs):
                            bj.vt_recv[i].copy_(bi.vt_send[j]); bj.qk_recv[:, :, i].copy_(bi.qk_send[:, :, j]); bj.o_recv[:, i].copy_(bi.o_send[:, j])
            cp1, cp2 = timeit(lambda: copies(1)), timeit(lambda: copies(2))
            r_two, r_pair = (ms2 - 2 * cp1) / P, (msp - cp2) / P
            print(f"P={P} G={G}: per rank compute + unpack per STEP: two shard forwards {r_two:.1f} ms ({r_two / (pair_base / P) - 1:+.1%} over the ideal 1/P of the single-rank "
                  f"stacked pair), stacked pair on the shard {r_pair:.1f} ms ({r_pair / (pair_base / P) - 1:+.1%}); simulated transport {2 * cp1:.1f} / {cp2:.1f} ms")
            if "tags" in sys.argv[1:]:
                tg = tags(lambda: sp.forward_local_pair(hs, x, t, ctx, ctx2, groups=G))
                print("      per tag, all shards (stacked) / single-rank stacked pair: " + "  ".join(f"{k} {tg.get(k, 0.0):.2f}/{v:.2f} ({tg.get(k, 0.0) / v - 1:+.0%})" for k, v in pair_tags.items()))
        del hs
    sys.exit(0)
for P in (2, 4, 6):
    hs = [handle() for _ in range(P)]
    for m in hs: m.context_cache(True)
    for G in sorted({1, sp.head_groups(12 // P, 32760)}):
        ms = timeit(lambda: sp.forward_local(hs, x, t, ctx, groups=G))
        # the simulated transport: the same device copies forward_local makes between the shards' buffers, timed alone
        bufs = [m._sp_buffers[k] for m in hs for k in m._sp_buffers if k[3] == G]
        def copies():
            for _ in range(layers):
                for j, bj in enumerate(bufs):
                    for i, bi in enumerate(bufs):
                        bj.vt_recv[i].copy_(bi.vt_send[j]); bj.qk_recv[:, :, i].copy_(bi.qk_send[:, :, j]); bj.o_recv[:, i].copy_(bi.o_send[:, j])
        cp = timeit(copies)
        per_rank = (ms - cp) / P
        print(f"P={P} G={G}: all shards back to back {ms:.1f} ms, of which simulated transport {cp:.1f} ms -> per rank compute + unpack {per_rank:.1f} ms "
              f"({per_rank / (base / P) - 1:+.1%} over an ideal 1/P split); exchange volume per block and rank {(32760 // P) * (1536 // P) * (P - 1) * 2 * 4 / 1e6:.1f} MB")
        if "tags" in sys.argv[1:]:
            tg = tags(lambda: sp.forward_local(hs, x, t, ctx, groups=G))
            print("      per tag, all shards / plain: " + "  ".join(f"{k} {tg.get(k, 0.0):.2f}/{v:.2f} ({tg.get(k, 0.0) / v - 1:+.0%})" for k, v in base_tags.items())
                  + f"  | tagged sum {sum(tg.values()):.1f}/{sum(base_tags.values()):.1f} of {ms:.1f}/{base:.1f} ms")
    del hs
