"""Captures the call signatures of the reference's public API for the hot path
FROM THE REFERENCE ITSELF into api_signatures.json.  Build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_api_signatures.py

Only names, parameter names, parameter order and printable defaults are recorded (data, not
source); tests/test_alias_cpu.py checks the ``fourier_feature_nets`` alias package against it.
"""

import inspect
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_goldens import REFERENCE, _install_stubs  # noqa: E402

# symbol -> methods whose signatures the three target scripts rely on
TABLE = {
    "NeRF": ["__init__", "forward", "save"],
    "FourierFeatureMLP": ["__init__", "forward", "save"],
    "MLP": ["__init__"],
    "BasicFourierMLP": ["__init__"],
    "PositionalFourierMLP": ["__init__"],
    "GaussianFourierMLP": ["__init__"],
    "RaySampler": ["__init__", "sample", "rays_for_camera", "to_valid", "to_image", "__len__"],
    "RaySamples": ["to", "pin_memory", "subset", "numpy"],
    "Raycaster": ["__init__", "render", "batched_render", "render_image", "fit"],
    "ImageDataset": ["__init__", "load", "get_rays", "loss", "render", "rays_for_camera",
                     "to_image", "sample_cameras", "subset", "to_valid", "index_for_camera",
                     "__len__"],
    "CameraInfo": ["create", "raycast"],
    "Resolution": ["scale_to_height", "square"],
    "RenderResult": ["to", "numpy"],
    "Voxels": ["__init__", "forward", "save"],
}
FUNCTIONS = ["load_model", "calculate_blend_weights", "orbit", "exponential_lr_decay", "linspace"]
FIELDS = {"RaySamples": "_fields", "RenderResult": "_fields", "CameraInfo": "_fields",
          "Resolution": "_fields", "LogEntry": "_fields"}


def describe(fn):
    out = []
    for p in inspect.signature(fn).parameters.values():
        if p.name in ("self", "cls"):
            continue
        default = None if p.default is inspect.Parameter.empty else repr(p.default)
        out.append({"name": p.name, "kind": p.kind.name, "default": default})
    return out


def main():
    _install_stubs()
    sys.path.insert(0, REFERENCE)
    import fourier_feature_nets as ref
    from fourier_feature_nets import ray_caster, utils

    def find(name):          # top-level export, else the submodule that defines it
        for owner in (ref, utils, ray_caster):
            if hasattr(owner, name):
                return getattr(owner, name)
        raise AttributeError(name)

    blob = {"classes": {}, "functions": {}, "fields": {}, "modes": []}
    for cls_name, methods in TABLE.items():
        cls = find(cls_name)
        blob["classes"][cls_name] = {m: describe(getattr(cls, m)) for m in methods}
    for name in FUNCTIONS:
        blob["functions"][name] = describe(find(name))
    for name, attr in FIELDS.items():
        blob["fields"][name] = list(getattr(find(name), attr))
    blob["modes"] = [m.name for m in ref.RayDataset.Mode]
    blob["exports"] = sorted(n for n in ref.__all__ if hasattr(ref, n))
    with open(os.path.join(HERE, "api_signatures.json"), "w") as f:
        json.dump(blob, f, indent=1, sort_keys=True)
    print("wrote api_signatures.json:", len(blob["classes"]), "classes")


if __name__ == "__main__":
    main()
