"""tools/pmc_to_json.py <tag>  — gpurun_out/<tag>_{flash,gemm_ffn1}_pmc.txt (tools/pmc_summary.py) -> gpurun_out/<tag>_{flash,gemm_ffn1}_pmc.json.

Per launch of the kernel: the raw counters, the HBM-side bytes corrected as MI355X_MICROARCH.md §HBM prescribes (FETCH_SIZE / WRITE_SIZE are KiB;
FETCH_SIZE tallies a wide coalesced read at half its bytes on gfx950 -> x2), the L2 hit rate, and the box-independent matrix-pipe figure
    mfma_busy_in_clock = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs)
i.e. the share of the kernel's own shader clocks in which a SIMD's matrix pipe was busy, whatever clock the power cap allowed.
Every JSON carries the sha256[:16] of every csrc/*.hip and of csrc/svi_common.h it was collected on; bench.py quotes it only when all of them equal the tree it times.
"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRCS = ["svi_attention.hip", "svi_gemm.hip", "svi_dit.hip", "svi_elementwise.hip", "svi_vae.hip", "svi_api.hip", "svi_encoders.hip",
        "svi_common.h"]          # the shared device code (MFMA fragment helpers, the MX quantiser, GELU) is part of every kernel: gated too


def source_hashes():
    return {s: hashlib.sha256(open(os.path.join(ROOT, "stable-video-infinity_amd", "csrc", s), "rb").read()).hexdigest()[:16] for s in SRCS}


def parse(path):
    """{kernel name: {counter: per-dispatch value}}"""
    out, cur = {}, None
    for line in open(path):
        if line.strip() and not line[0].isspace():
            cur = line.strip()
            out[cur] = {}
            continue
        m = re.match(r"\s+(\S+)\s+per-dispatch\s+([0-9.]+)", line)
        if m and cur:
            out[cur][m.group(1)] = float(m.group(2))
    return out


def derive(vals):
    d = {}
    f, w = vals.get("FETCH_SIZE"), vals.get("WRITE_SIZE")
    d["fetch_bytes"] = None if f is None else f * 1024 * 2
    d["write_bytes"] = None if w is None else w * 1024
    d["hbm_bytes"] = None if f is None or w is None else d["fetch_bytes"] + d["write_bytes"]
    if vals.get("SQ_VALU_MFMA_BUSY_CYCLES") and vals.get("GRBM_GUI_ACTIVE"):
        d["mfma_busy_in_clock"] = round((vals["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (vals["GRBM_GUI_ACTIVE"] / 8.0), 4)
    if vals.get("TCC_HIT_sum") is not None and vals.get("TCC_MISS_sum") is not None and vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"] > 0:
        d["l2_hit_rate"] = round(vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]), 4)
    if vals.get("SQ_WAVE_CYCLES"):
        for k, name in (("SQ_WAIT_INST_ANY", "wait_inst_share"), ("SQ_WAIT_ANY", "wait_any_share")):
            if vals.get(k) is not None:
                d[name] = round(vals[k] / vals["SQ_WAVE_CYCLES"], 4)
    return d


def main(tag):
    note = ("FETCH_SIZE/WRITE_SIZE are KiB; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md HBM section: wide coalesced reads are tallied at half); "
            "separate --pmc passes, tools/pmc_collect.sh")
    hashes = source_hashes()
    # self-attention: one call = optimistic pass + flagged second pass (two instantiations of flash_fwd2_kernel): their counters are added
    p = os.path.join(ROOT, "gpurun_out", f"{tag}_flash_pmc.txt")
    if os.path.exists(p):
        ks = {k: v for k, v in parse(p).items() if "flash_fwd2_kernel" in k or "flash_fwd3_kernel" in k}
        vals = {}
        for v in ks.values():
            for c, x in v.items():
                vals[c] = vals.get(c, 0.0) + x
        out = {"kernel": "flash_fwd3_kernel + flash_fwd2_kernel<.., 2> (self-attention, L=32760, 12 heads; optimistic pass on v_mfma_f32_16x16x32_bf16 + flagged second pass)", "kernels_summed": sorted(ks),
               "counters_per_launch": vals, "attention_src_sha": hashes["svi_attention.hip"], "source_hashes": hashes, **derive(vals), "note": note}
        json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"{tag}_flash_pmc.json"), "w"), indent=1)
        print(json.dumps({k: out.get(k) for k in ("hbm_bytes", "mfma_busy_in_clock", "l2_hit_rate")}))
    # the opt-in fp8 QK^T attention (SVI_ATTN_QK8=1): the same probe; the two quantiser launches are listed beside the attention kernels
    p = os.path.join(ROOT, "gpurun_out", f"{tag}_flash_qk8_pmc.txt")
    if os.path.exists(p):
        allk = parse(p)
        ks = {k: v for k, v in allk.items() if "flash_fwd2_kernel" in k}
        vals = {}
        for v in ks.values():
            for c, x in v.items():
                vals[c] = vals.get(c, 0.0) + x
        qz = {k: v for k, v in allk.items() if "mx8_quantize" in k}
        out = {"kernel": "flash_fwd2_kernel<QK8> (opt-in SVI_ATTN_QK8: QK^T on v_mfma_scale_f32_32x32x64_f8f6f4, P·V bf16; L=32760, 12 heads; both passes)",
               "kernels_summed": sorted(ks), "counters_per_launch": vals, "quantiser_counters_per_dispatch": qz, "source_hashes": hashes, **derive(vals), "note": note}
        json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"{tag}_flash_qk8_pmc.json"), "w"), indent=1)
        print(json.dumps({k: out.get(k) for k in ("kernel", "hbm_bytes", "mfma_busy_in_clock", "l2_hit_rate")}))
    for key, what, alg in (("gemm_ffn1", "ffn1: M = 32760, N = 8960, K = 1536, GELU-tanh epilogue", (32760 * 1536 + 8960 * 1536 + 32760 * 8960) * 2),
                           ("gemm_ffn2", "ffn2: M = 32760, N = 1536, K = 8960, gate + residual epilogue", (32760 * 8960 + 1536 * 8960 + 2 * 32760 * 1536) * 2)):
        p = os.path.join(ROOT, "gpurun_out", f"{tag}_{key}_pmc.txt")
        if not os.path.exists(p):
            continue
        ks = {k: v for k, v in parse(p).items() if "gemm" in k and v.get("SQ_INSTS_MFMA")}
        if ks:
            name = max(ks, key=lambda k: ks[k]["SQ_INSTS_MFMA"])          # the GEMM launch itself
            vals = ks[name]
            out = {"kernel": f"{name} ({what})", "counters_per_launch": vals, "source_hashes": hashes, **derive(vals), "algorithmic_bytes": alg, "note": note}
            json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"{tag}_{key}_pmc.json"), "w"), indent=1)
            print(json.dumps({k: out.get(k) for k in ("kernel", "hbm_bytes", "mfma_busy_in_clock", "l2_hit_rate")}))
    # the fused cross-attention (flash_cross_resident_kernel: q read + o written, HBM-bound) and the VAE's strip convolution, when their passes were collected
    for key, pat, what in (("flash_cross", "flash_cross_resident_kernel", "cross-attention, L = 32760 query rows, 12 heads, 65 keys resident in LDS, q RMS-normalised as it is read"),
                           ("vae_conv", "conv_dma2h_pair_kernel", "the VAE decoder's residual-block convolution (fp16 two-term, plane-fed strips, two tiles per workgroup)")):
        p = os.path.join(ROOT, "gpurun_out", f"{tag}_{key}_pmc.txt")
        if not os.path.exists(p):
            continue
        ks = {k: v for k, v in parse(p).items() if pat in k and v.get("GRBM_GUI_ACTIVE")}
        if ks:
            name = max(ks, key=lambda k: ks[k].get("SQ_WAVE_CYCLES", 0.0))
            vals = ks[name]
            out = {"kernel": f"{name} ({what})", "counters_per_launch": vals, "source_hashes": hashes, **derive(vals), "note": note}
            json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"{tag}_{key}_pmc.json"), "w"), indent=1)
            print(json.dumps({k: out.get(k) for k in ("kernel", "hbm_bytes", "mfma_busy_in_clock", "l2_hit_rate")}))
    json.dump(hashes, open(os.path.join(ROOT, "gpurun_out", f"{tag}_source_hashes.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])
