"""Which E8M0 scale does v_mfma_scale_f32_32x32x64_f8f6f4 apply to which K octet, as svi_gemm_mx8 feeds it?  All activations 1.0, block
scales 2^0, 2^1, 2^2, 2^3 over the four 32-element blocks of one 128-wide K tile, weight row n = ones on K octet n only:
out[m][n] = 8 * (scale applied to octet n).  With the operand arrangement of csrc/svi_gemm.hip: 8 8 8 8 16 16 16 16 32 32 32 32 64 64 64 64.
(With a lane holding 32 CONSECUTIVE K elements the answer was 8 8 16 16 8 8 16 16 | 32 32 64 64 32 32 64 64: the hardware applies the first
block's scale to every lane's first 16 bytes and the second block's to its last 16 — which is how the layout in the kernel was found.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd"))
import torch
from svi_hip import _lib as L
M, N, K = 256, 256, 128
q = torch.full((M, K), 0x38, dtype=torch.uint8, device="cuda")                  # e4m3 1.0
tab = torch.full((1, 256), (127 | (128 << 8) | (129 << 16) | (130 << 24)) - (1 << 32), dtype=torch.int32, device="cuda")
w = torch.zeros((N, K), dtype=torch.uint8, device="cuda")
for n in range(16):
    w[n, 8 * n:8 * n + 8] = 0x38
out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
L.check(L.lib().svi_gemm_mx8(q.data_ptr(), K, tab.data_ptr(), 256, w.data_ptr(), K, out.data_ptr(), N, M, N, K, None, 0, None, None, N, L.current_stream()))
torch.cuda.synchronize()
print("row 0  :", [float(v) for v in out[0, :16].float()])
print("row 37 :", [float(v) for v in out[37, :16].float()])
print("row 200:", [float(v) for v in out[200, :16].float()])
print("all rows equal:", bool((out[:, :16] == out[0:1, :16]).all()))
# second probe: activations differ per K octet (value 2^-j on octet j... e4m3 1.0 * 2^-? ) with UNIT scales: checks the A/B pairing of k
q2 = torch.empty((M, K), dtype=torch.uint8, device="cuda")
vals = [0x38, 0x30, 0x28, 0x20, 0x40, 0x48, 0x50, 0x58] * 2                       # 1, .5, .25, .125, 2, 4, 8, 16
for j in range(16):
    q2[:, 8 * j:8 * j + 8] = vals[j]
tab1 = torch.full((1, 256), 0x7f7f7f7f, dtype=torch.int32, device="cuda")
L.check(L.lib().svi_gemm_mx8(q2.data_ptr(), K, tab1.data_ptr(), 256, w.data_ptr(), K, out.data_ptr(), N, M, N, K, None, 0, None, None, N, L.current_stream()))
torch.cuda.synchronize()
print("pairing:", [float(v) for v in out[5, :16].float()], "(expected 8, 4, 2, 1, 16, 32, 64, 128, twice)")
