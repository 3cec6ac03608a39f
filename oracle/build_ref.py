"""Recipe for `oracle/_ref/` — the REFERENCE's own modules for this path, made available to the checker — TEST INFRASTRUCTURE ONLY.

The reference is Python, so there is nothing to compile: "building" `oracle/_ref` means copying, at build time and only when
`/root/reference` is present (the build container), the six reference files the import shim of SURVEY.md Appendix A needs into
`oracle/_ref/diffsynth/...`:

    models/wan_video_dit.py    DiTBlock / WanModel                    (imports the next three at module level)
    models/attention.py        flash_attention dispatch               (wan_video_dit.py:26)
    models/utils.py            hash_state_dict_keys                   (wan_video_dit.py:7)
    utils/multitalk_utils.py   get_attn_map_with_target               (wan_video_dit.py:27)
    models/wan_video_vae.py    WanVideoVAE
    schedulers/flow_match.py   FlowMatchScheduler

`oracle/_ref/` is listed in .gitignore — reference sources never enter this repository's history — but not in .gpurunignore, so the
copies travel to the GPU box next to the built .so files.  There `bench.py`'s `cpu_baseline` child times the reference's own
`DiTBlock.forward` and `WanVideoVAE.decode` on the box's host cores (`cpu_baseline.kind = "reference"`); without `oracle/_ref` it
times the restatement (`oracle/wan_dit_oracle.py`, kind "port").  Nothing in the product package imports this directory.

    python oracle/build_ref.py            # (re)creates oracle/_ref from /root/reference; a no-op where the reference is absent
"""
from __future__ import annotations

import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
FILES = ["models/wan_video_dit.py", "models/attention.py", "models/utils.py", "utils/multitalk_utils.py", "models/wan_video_vae.py",
         "schedulers/flow_match.py"]


def build(reference: str = os.environ.get("SVI_REFERENCE", "/root/reference"), out: str = OUT) -> bool:
    """True when oracle/_ref holds the reference modules afterwards (copied now, or already there from the build container)."""
    src = os.path.join(reference, "diffsynth")
    if not os.path.isdir(src):
        return os.path.isfile(os.path.join(out, "MANIFEST.json"))
    manifest = {}
    for rel in FILES:
        dst = os.path.join(out, "diffsynth", rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copyfile(os.path.join(src, rel), dst)
        manifest["diffsynth/" + rel] = hashlib.sha256(open(dst, "rb").read()).hexdigest()
    json.dump({"source": reference, "files": manifest}, open(os.path.join(out, "MANIFEST.json"), "w"), indent=1)
    return True


def available(out: str = OUT) -> bool:
    return os.path.isfile(os.path.join(out, "MANIFEST.json")) and all(os.path.isfile(os.path.join(out, "diffsynth", f)) for f in FILES)


if __name__ == "__main__":
    ok = build()
    print(f"oracle/_ref: {'ready' if ok else 'reference absent, nothing copied'}")
    sys.exit(0)
