"""Is the VAE decode held by the power limit?  Shader clock / socket power (rocm-smi) sampled while the C2-size decode loops.   python tools/vae_power.py [decodes]"""
import os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from svi_hip.vae import WanVideoVAE, device_vae_weights
dev = torch.device("cuda")
vae = WanVideoVAE.from_state_dict(device_vae_weights(0, dev))
z = torch.randn((16, 21, 60, 104), generator=torch.Generator(device=dev).manual_seed(3), device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
samples, stop = [], False
def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
        pw = [l.split(":")[-1].strip() for l in out.splitlines() if "Power" in l]
        sc = [l.split("(")[-1].split(")")[0] for l in out.splitlines() if "sclk" in l]
        samples.append((time.time(), pw[:1], sc[:1])); time.sleep(0.2)
vae.decode([z], device=dev); torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
t0 = time.time()
for _ in range(n):
    vae.decode([z], device=dev)
torch.cuda.synchronize()
t1 = time.time(); stop = True; th.join()
print(f"{n} decodes, {(t1 - t0) / n * 1e3:.1f} ms each")
for t, p, c in samples: print(f"  t+{t - t0:5.2f}s  power {p}  sclk {c}")
