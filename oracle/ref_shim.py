"""Import the reference's own wan_video_dit / wan_video_vae / flow_match modules from a directory tree WITHOUT `import diffsynth`
(which fails here: its __init__ pulls the whole model zoo and absent third-party packages) — TEST INFRASTRUCTURE ONLY.

The recipe is SURVEY.md Appendix A: `diffsynth`, `diffsynth.models`, `diffsynth.utils`, `diffsynth.schedulers` are registered as empty
namespace packages that point at `<root>/diffsynth/...`, and the third-party modules that only import lines touch are stubbed.
`root` is `/root/reference` in the build container (tests/gen_golden.py) or `oracle/_ref` (oracle/build_ref.py's copy) on the GPU box.
"""
from __future__ import annotations

import importlib
import importlib.machinery
import sys
import types


def load(root: str):
    """-> (wan_video_dit, wan_video_vae, flow_match) modules of the reference tree under `root`."""
    import torch

    def ns(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
        sys.modules[name] = m

    for pkg in ("diffsynth", "diffsynth.models", "diffsynth.utils", "diffsynth.schedulers"):
        ns(pkg, root + "/" + pkg.replace(".", "/"))

    class Stub(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return type(k, (object,), {})

    for n in ("diffusers", "diffusers.configuration_utils", "xfuser", "xfuser.core", "xfuser.core.distributed",
              "xformers", "xformers.ops", "imageio", "torchvision", "torchvision.transforms"):
        sys.modules.setdefault(n, Stub(n))
    sys.modules["diffusers.configuration_utils"].register_to_config = lambda f: f
    # AudioProjModel(ModelMixin, ConfigMixin) (wan_video_dit.py:44) must be a real nn.Module for its parameters to register
    if isinstance(sys.modules["diffusers"], Stub):
        sys.modules["diffusers"].ModelMixin = torch.nn.Module
        sys.modules["diffusers.configuration_utils"].ConfigMixin = type("ConfigMixin", (object,), {})
    dit = importlib.import_module("diffsynth.models.wan_video_dit")
    vae = importlib.import_module("diffsynth.models.wan_video_vae")
    fm = importlib.import_module("diffsynth.schedulers.flow_match")
    return dit, vae, fm
