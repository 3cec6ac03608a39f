"""Audit the hand-owned register contract of flash_fwd2_kernel (csrc/svi_attention.hip) in the emitted gfx950 assembly.

    python tools/audit_flash2.py
Compiles svi_attention.hip with -save-temps into a scratch dir and checks, for every flash_fwd2_kernel instantiation:
  * hipcc-generated v_accvgpr_read/write (printed `aN`, our inline asm prints `a[N]`) only touch a0..a63 (a0..a95 in the QK8
    instantiations, whose owned registers start 32 higher);
  * no scratch (private segment) access between the loop header and the loop's last back-edge;
  * reports VGPR/AGPR/scratch/spill figures and the per-tile instruction mix of the steady loop.
Exit status 1 on a violation.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "stable-video-infinity_amd", "csrc", "svi_attention.hip")
tmp = tempfile.mkdtemp(prefix="audit_flash2_")
OUT = os.path.join(tmp, "svi_attention-hip-amdgcn-amd-amdhsa-gfx950.s")
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--cuda-device-only", "-S", SRC, "-o", OUT]
subprocess.run(cmd, check=True, cwd=tmp, capture_output=True)
asm = open(OUT).read().splitlines()
bad = 0
i = 0
while i < len(asm):
    m = re.match(r"^(_Z17flash_fwd2_kernel\w+):", asm[i])
    if not m:
        i += 1
        continue
    name = m.group(1)
    j = i
    while "s_endpgm" not in asm[j]:
        j += 1
    body = asm[i:j]
    comp = [l for l in body if "accvgpr" in l and "a[" not in l]
    idx = [int(x) for l in comp for x in re.findall(r"\ba(\d+)\b", l)]
    hdr = [k for k, l in enumerate(body) if "Loop Header" in l]
    loop_scratch = 0
    mix = {}
    if hdr:
        start = hdr[0]
        # the loop's label is the last label before the header comment; its last back-edge closes the loop
        lab = None
        for k in range(start, -1, -1):
            mm = re.match(r"^(\.LBB\d+_\d+):", body[k])
            if mm:
                lab = mm.group(1)
                break
        ends = [k for k, l in enumerate(body) if lab and re.search(r"s_c?branch\S*\s+" + re.escape(lab) + r"\b", l)]
        end = ends[-1] if ends else len(body)
        first = min([k for k, l in enumerate(body) if lab and l.startswith(lab + ":")] + [start])
        for l in body[first:end]:
            t = l.split()
            if not t or t[0].startswith(";") or t[0].endswith(":"):
                continue
            mix[t[0]] = mix.get(t[0], 0) + 1
            if t[0].startswith("scratch_"):
                loop_scratch += 1
    meta = {}
    for l in asm:
        pass
    print(f"{name}: compiler AGPR copies {len(comp)} (max index {max(idx) if idx else '-'}), scratch ops in loop {loop_scratch}")
    if mix:
        top = sorted(mix.items(), key=lambda kv: -kv[1])[:14]
        print("   loop mix (2 tiles): " + ", ".join(f"{k} {v}" for k, v in top))
    own0 = 96 if re.search(r"Lb1EEv", name) else 64          # last template argument: QK8
    if idx and max(idx) >= own0:
        print(f"   VIOLATION: hipcc uses an AGPR >= a{own0} (owned by the kernel)")
        bad = 1
    if loop_scratch:
        print("   VIOLATION: scratch access inside the tile loop")
        bad = 1
    i = j
sys.exit(bad)
