"""What the built library's gfx950 code objects say about the hot kernels, without a GPU: the register budgets the measured numbers rest on.

A kernel that starts to spill (a hoisted loop invariant, one more live fragment) keeps every parity test green and quietly loses its roofline
fraction — this round's resident cross-attention kernel did exactly that twice while it was being written.  The kernel metadata of the embedded
code objects (`llvm-objdump --offloading` + `llvm-readelf --notes`: `.private_segment_fixed_size`, `.vgpr_count`, `.vgpr_spill_count`) is read from
the very libsvi_hip.so the GPU tests load."""
import os
import re
import shutil
import subprocess

import pytest

from svi_hip import _lib as L

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_table(tmp_path):
    objdump, readelf = os.path.join(LLVM, "llvm-objdump"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(objdump) and os.path.exists(readelf) and shutil.which("c++filt")):
        pytest.skip("llvm-objdump / llvm-readelf / c++filt not in this image")
    so = tmp_path / "libsvi_hip.so"
    shutil.copy(L.LIB_PATH, so)                 # --offloading writes the bundles beside its input: keep them out of the tree
    subprocess.run([objdump, "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)
    table = {}
    for f in sorted(tmp_path.glob("libsvi_hip.so.*gfx950")):
        notes = subprocess.run([readelf, "--notes", str(f)], check=True, capture_output=True, text=True).stdout
        name = None
        for line in notes.splitlines():
            m = re.match(r"\s*\.(name|private_segment_fixed_size|vgpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size):\s+(\S+)", line)
            if not m:
                continue
            if m.group(1) == "name":
                name = m.group(2)
                table[name] = {}
            elif name:
                table[name][m.group(1)] = int(m.group(2))
    names = list(table)
    plain = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return {re.sub(r"\(.*", "", p.replace("(anonymous namespace)::", "").replace("void ", "")): table[n] for n, p in zip(names, plain)}


def test_hot_kernels_have_no_scratch_and_fit_their_occupancy(tmp_path):
    """(a) every kernel a default C2 step or the VAE decode launches runs without scratch memory and without spilled registers; (b) the occupancy each
    one is written for holds: two waves per SIMD (<= 256 VGPRs) for the tiled GEMM and the resident cross-attention, ONE wave per SIMD owning the
    whole 512-entry file for the self-attention kernel's optimistic pass (the pass that is 64 % of a step).  The rarely taken passes of the
    self-attention kernel (single-pass with rescaling, the flagged second pass) are allowed their 8 spilled registers — they are listed, not hidden."""
    t = kernel_table(tmp_path)
    assert len(t) > 100, len(t)

    def family(prefix):
        got = {k: v for k, v in t.items() if k.startswith(prefix)}
        assert got, f"no kernel named {prefix}* in the library"
        return got
    clean = {}
    for prefix in ("flash_cross_resident_kernel<", "gemm_bf16_nt_256e_kernel<", "gemm_bf16_nt_kernel", "gemm_mx8_nt_256_kernel", "ln_mod_rows_kernel<", "rmsnorm_rope_rows_kernel<",
                   "row_rs_kernel", "cfg_step_kernel", "conv_dma2h_kernel<", "flash_fwd_kernel<"):
        clean.update(family(prefix))
    main_pass = {k: v for k, v in family("flash_fwd2_kernel<").items() if re.match(r"flash_fwd2_kernel<\d+, \d+, (true|false), 1, ", k)}
    assert len(main_pass) >= 4
    main_pass.update(family("flash_fwd3_kernel<"))          # round 6: the optimistic pass on v_mfma_f32_16x16x32_bf16 (what the step launches by default)
    assert len(main_pass) >= 8
    clean.update(main_pass)
    # (scalar registers parked in the lanes of a vector register — sgpr_spill_count, a few in the convolution and the Q8 attention — cost no memory)
    bad = {k: v for k, v in clean.items() if v["private_segment_fixed_size"] or v["vgpr_spill_count"]}
    assert not bad, bad
    for k, v in family("flash_cross_resident_kernel<").items():
        assert v["vgpr_count"] <= 256 and v["group_segment_fixed_size"] == 0, (k, v)          # two waves per SIMD; LDS is dynamic (128 KiB + the mask table)
    for k, v in family("gemm_bf16_nt_256e_kernel<").items():
        assert v["vgpr_count"] <= 256, (k, v)
    # round 6: the 128^2 tile's loops — <1, 256> shares a SIMD with a second workgroup, <4, 512> (the C1-size step's projections: at most one workgroup per CU)
    # brings its own second wave per SIMD, <4, 256> (A/B only) owns the file
    g128 = family("gemm_bf16_nt_kernel<")
    assert {re.sub(r"\s", "", k) for k in g128} >= {"gemm_bf16_nt_kernel<1,256>", "gemm_bf16_nt_kernel<4,256>", "gemm_bf16_nt_kernel<4,512>"}, sorted(g128)
    for k, v in g128.items():
        assert v["vgpr_count"] <= (512 if re.sub(r"\s", "", k).endswith("<4,256>") else 256), (k, v)
    for k, v in main_pass.items():
        assert 256 < v["vgpr_count"] <= 512, (k, v)                                             # one wave per SIMD, by design
    other_passes = {k: v for k, v in family("flash_fwd2_kernel<").items() if k not in main_pass}
    assert all(v["private_segment_fixed_size"] <= 64 for v in other_passes.values()), other_passes
