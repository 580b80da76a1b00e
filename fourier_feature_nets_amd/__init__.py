"""MI355X-native NeRF volume rendering behind the ``fourier_feature_nets`` class surface.

Everything numeric runs in hand-written gfx950 kernels (``csrc/``) reached through the C ABI
of ``include/ffn_hip.h``; this package is the host-side mirror of the reference's Python API
for that path.  There is no CPU fallback: using a model or sampler without a GPU raises.
"""

from .cameras import CameraInfo, Resolution, orbit
from .caster import LogEntry, Raycaster, TrainEngine
from .dataset import ImageDataset, RayDataset
from .frames import FrameSink
from .occupancy import OccupancyGrid
from .models import (
    BasicFourierMLP,
    FourierFeatureMLP,
    GaussianFourierMLP,
    MLP,
    NeRF,
    PositionalFourierMLP,
)
from .sampler import RaySampler, RaySamples
from .utils import (
    ETABar,
    RenderResult,
    calculate_blend_weights,
    exponential_lr_decay,
    linspace,
    load_model,
)
from .visualizers import ActivationVisualizer, EvaluationVisualizer, OrbitVideoVisualizer, Visualizer
from .voxels import Voxels

__version__ = "0.1.0"

__all__ = ["__version__", "ActivationVisualizer", "BasicFourierMLP", "CameraInfo", "ETABar", "EvaluationVisualizer", "FourierFeatureMLP", "FrameSink",
           "GaussianFourierMLP", "ImageDataset", "LogEntry", "MLP", "NeRF",
           "OccupancyGrid", "OrbitVideoVisualizer", "PositionalFourierMLP", "RayDataset", "RaySampler", "RaySamples", "Raycaster",
           "RenderResult", "Resolution", "TrainEngine", "Visualizer", "Voxels", "calculate_blend_weights",
           "exponential_lr_decay", "linspace", "load_model", "orbit"]
