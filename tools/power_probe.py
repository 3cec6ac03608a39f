"""Sample clocks / power (rocm-smi) while one kernel runs in a loop: is the sustained MFMA rate power-limited?
    python tools/power_probe.py attn|gemm|gemmx [seconds]
"""
import os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd"))
import torch
import svi_hip
from svi_hip import _lib as L
what = sys.argv[1] if len(sys.argv) > 1 else "attn"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
samples = []
stop = False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=10).stdout
            keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("Power", "sclk", "mclk", "junction", "fclk"))]
            samples.append((time.time(), keep))
        except Exception as ex:
            samples.append((time.time(), [repr(ex)]))
        time.sleep(0.3)
if what == "attn":
    n = 12; s = 32760
    q, k, v = [(torch.randn((1, s, n * 128), generator=g, device=dev)).to(torch.bfloat16) for _ in range(3)]
    fl = 4.0 * s * s * n * 128
    run = lambda: svi_hip.flash_attention(q, k, v, n)
else:
    M = N = K = 8192
    x = (torch.randn((M, K), generator=g, device=dev)).to(torch.bfloat16); w = (torch.randn((N, K), generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    if what == "gemmz":
        x.zero_(); w.zero_()
    b = torch.zeros(N, dtype=torch.bfloat16, device=dev); out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    lib = L.lib(); st = L.current_stream()
    fl = 2.0 * M * N * K
    run = lambda: L.check(lib.svi_gemm_bf16(x.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, b.data_ptr(), 0, L.EPI_BIAS, None, None, N, st))
run(); torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); iters = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.time() - t0 < secs:
    for _ in range(20):
        run()
    iters += 20
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
stop = True; th.join()
ms = e0.elapsed_time(e1) / iters
print(f"{what}: {ms:.3f} ms/launch, {fl/ms/1e9:.0f} TFLOP/s sustained over {secs}s")
for t, keep in samples[:: max(1, len(samples) // 6)]:
    print(f"  t+{t-t0:4.1f}s  " + " | ".join(keep)[:400])
