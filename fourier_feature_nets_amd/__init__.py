"""MI355X-native NeRF volume rendering behind the ``fourier_feature_nets`` class surface.

Everything numeric runs in hand-written gfx950 kernels (``csrc/``) reached through the C ABI
of ``include/ffn_hip.h``; this package is the host-side mirror of the reference's Python API
for that path.  There is no CPU fallback: using a model or sampler without a GPU raises.
"""

from .models import (
    BasicFourierMLP,
    FourierFeatureMLP,
    GaussianFourierMLP,
    MLP,
    NeRF,
    PositionalFourierMLP,
)

__version__ = "0.1.0"
