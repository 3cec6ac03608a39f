"""Build (here) / run (on the GPU box) the MFMA issue-shadow probe.  python tools/probe/run_mfma_issue_probe.py [build]"""
import ctypes as C, os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libmfma_issue_probe.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", os.path.join(HERE, "mfma_issue_probe.hip"), "-o", SO], check=True)
    print("built", SO); sys.exit(0)
lib = C.CDLL(SO)
NAMES = {0: "V4 +0", 1: "V4 +1 fma", 2: "V4 +2 fma", 3: "V4 +3 fma", 4: "V4 +4 fma", 5: "V4 +5 fma", 6: "V4 +6 fma", 7: "V4 +7 fma", 8: "V4 +8 fma",
         10: "A4 +0", 11: "A4 +2 fma", 12: "A4 +4 fma", 13: "A4 +5 fma", 14: "A4 +6 fma", 15: "A4 +8 fma",
         20: "V2 +0", 21: "V2 +2 fma", 22: "V2 +4 fma", 23: "V2 +6 fma",
         30: "V4 +1 exp", 31: "V4 +2 exp", 32: "V4 +3 exp", 33: "V4 +4 exp", 34: "A4 +2 exp", 35: "A4 +4 exp",
         40: "V4 +2 max3", 41: "V4 +4 max3", 42: "V4 +2 cvt_pk", 43: "V4 +4 cvt_pk", 44: "V4 +4 s_nop", 45: "V4 +8 s_nop", 46: "V4 +4 add",
         50: "V4 + B5 (exp exp add add cvt)", 51: "V4 + B7 (fma fma exp exp add add cvt)", 52: "A4 + B5", 53: "V2 + B5"}
loops = 2000
out = (C.c_ulonglong * 1024)()
ms = C.c_float()
print("variant: accumulators(V=VGPR,A=AGPR; number of rotating chains) + fillers per MFMA -> cycles per MFMA (s_memtime), wall ns per MFMA")
for v in sorted(NAMES):
    rc = lib.probe_run(v, loops, out, C.byref(ms))
    if rc:
        print(v, "failed", rc); continue
    cyc = sorted(out[i] for i in range(1024))
    med = cyc[len(cyc) // 2] / (loops * 16.0)
    print(f"{v:3d} {NAMES[v]:36s} {med:7.1f} cyc/MFMA (memtime ticks)   {ms.value * 1e6 / (loops * 16):7.2f} ns/MFMA")
