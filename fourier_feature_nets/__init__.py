"""``import fourier_feature_nets as ffn`` on an MI355X box: the reference's package name bound to
the HIP implementation (``fourier_feature_nets_amd``), so that scripts written against the
reference (train_nerf.py:7, train_tiny_nerf.py:7, orbit_video.py:6) import unchanged.

Nothing lives here: every name, and every submodule already loaded, is the implementation's own
object (``fourier_feature_nets.utils is fourier_feature_nets_amd.utils``).  The reference's
module file names that differ from ours are mapped as well, for code that reaches into
submodules (``from fourier_feature_nets.ray_sampler import RaySamples``).
"""

import sys as _sys

import fourier_feature_nets_amd as _impl
from fourier_feature_nets_amd import *  # noqa: F401,F403
from fourier_feature_nets_amd import __version__  # noqa: F401

__all__ = list(_impl.__all__)

# reference module name -> implementation module that holds the same symbols
_SUBMODULES = {
    "utils": "utils", "visualizers": "visualizers", "camera_info": "cameras",
    "ray_sampler": "sampler", "ray_caster": "caster", "ray_dataset": "dataset",
    "image_dataset": "dataset", "fourier_feature_models": "models", "nerf_model": "models",
    "voxels_model": "voxels", "version": None,
}
for _ref_name, _ours in _SUBMODULES.items():
    _mod = _impl if _ours is None else _sys.modules[_impl.__name__ + "." + _ours]
    _sys.modules[__name__ + "." + _ref_name] = _mod
    globals()[_ref_name] = _mod
del _ref_name, _ours, _mod
