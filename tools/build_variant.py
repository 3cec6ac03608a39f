"""Build a variant of the library for same-box A/B runs:   python tools/build_variant.py NAME -DMACRO[=V] ...
-> stable-video-infinity_amd/svi_hip/libsvi_hip_NAME.so (objects under csrc/obj_NAME); select it with SVI_HIP_LIB=<path>."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-video-infinity_amd"))
from svi_hip import build as b
name, defs = sys.argv[1], sys.argv[2:]
obj = os.path.join(b.CSRC, "obj_" + name); os.makedirs(obj, exist_ok=True)
lib = os.path.join(b.HERE, f"libsvi_hip_{name}.so")
procs = []
for s in b.SOURCES:
    o = os.path.join(obj, s.replace(".hip", ".o"))
    procs.append((o, subprocess.Popen([b._hipcc(), *b.FLAGS, *defs, "-c", os.path.join(b.CSRC, s), "-o", o])))
for o, p in procs:
    assert p.wait() == 0, o
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *[o for o, _ in procs]])
print(lib)
