"""Import the reference hot path from /root/reference WITHOUT executing its package
__init__ files (they pull aim/torchvision/cleanfid, absent here). Recipe: SURVEY.md §8(c).

Only used in the build container by tools/make_goldens.py; never shipped to / run on the GPU box.
"""
import sys
import types

REF = "/root/reference"
sys.dont_write_bytecode = True  # never write __pycache__ into the read-only reference tree


def import_reference():
    for name in ("aim", "torchvision", "torchvision.utils", "torchvision.transforms"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["aim"].Image = object
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    for pkg in ("normalizing_flow", "diffusion_prior"):
        m = types.ModuleType(pkg)
        m.__path__ = [f"{REF}/{pkg}"]
        sys.modules[pkg] = m
    import importlib
    nf_utils = importlib.import_module("normalizing_flow.utils")
    sys.modules["normalizing_flow"].calculate_output_shapes = nf_utils.calculate_output_shapes
    mods = {
        "nf_utils": nf_utils,
        "transforms": importlib.import_module("normalizing_flow.transforms"),
        "glow": importlib.import_module("normalizing_flow.glow"),
        "prior": importlib.import_module("normalizing_flow.prior"),
        "gd": importlib.import_module("diffusion_prior.gaussian_diffusion"),
        "model": importlib.import_module("diffusion_prior.model"),
        "formaters": importlib.import_module("diffusion_prior.latent_formaters"),
    }
    return mods
