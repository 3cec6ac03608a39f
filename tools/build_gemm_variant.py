"""A variant of the library that differs in ONE source only (kernel experiments):   python tools/build_gemm_variant.py NAME [--src svi_vae.hip] -DMACRO[=V] ...
(default source: svi_gemm.hip) -> svi_hip/libsvi_hip_NAME.so, linked from the main build's other objects (run __graft_entry__.build first); select it with SVI_HIP_LIB=<path>."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd"))
from svi_hip import build as b
name, defs = sys.argv[1], sys.argv[2:]
src = "svi_gemm.hip"
if "--src" in defs:
    i = defs.index("--src"); src = defs[i + 1]; del defs[i:i + 2]
obj = os.path.join(b.CSRC, "obj_" + name); os.makedirs(obj, exist_ok=True)
o = os.path.join(obj, src.replace(".hip", ".o"))
subprocess.check_call([b._hipcc(), *b.FLAGS, *defs, "-c", os.path.join(b.CSRC, src), "-o", o])
others = [os.path.join(b.OBJ, s.replace(".hip", ".o")) for s in b.SOURCES if s != src]
lib = os.path.join(b.HERE, f"libsvi_hip_{name}.so")
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, o, *others])
print(lib)
