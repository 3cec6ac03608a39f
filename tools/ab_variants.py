"""Run one tool alternately under two builds of the library on the SAME box (box-to-box variance on this pool is up to 20 %).
    python tools/ab_variants.py NAME rounds -- python tools/attn_abl.py 3"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, rounds = sys.argv[1], int(sys.argv[2]); cmd = sys.argv[sys.argv.index("--") + 1:]
libs = {"base": os.path.join(ROOT, "stable-video-infinity_amd", "svi_hip", "libsvi_hip.so"),
        name: os.path.join(ROOT, "stable-video-infinity_amd", "svi_hip", f"libsvi_hip_{name}.so")}
for r in range(rounds):
    for k, lib in libs.items():
        out = subprocess.run(cmd, env=dict(os.environ, SVI_HIP_LIB=lib), capture_output=True, text=True).stdout
        for line in out.splitlines():
            if "amdgpu.ids" not in line:
                print(f"[{k:>8s} r{r}] {line}", flush=True)
