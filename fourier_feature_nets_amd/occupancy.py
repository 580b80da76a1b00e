"""Opt-in empty-space skipping for rendering.

The reference renders every sample of every ray; its only spatial acceleration structure is a
CPU octree used by the lecture visualisations (octree.py:418-501, voxelize_model.py:65-88 builds
one from depth maps).  Here an occupancy grid is built on the GPU from the density model itself
and ``Raycaster.render`` (inference only) runs the fused MLP on the samples that fall into
occupied cells.  Skipped samples get sigma = softplus(-100) = 0, so compositing is unchanged.
This is new behaviour: parity with the full render is PSNR-level (see tests), never used for
training, and off unless ``Raycaster.occupancy`` is set.
"""

from typing import Optional

import numpy as np
import torch

from . import ops


class OccupancyGrid:
    """resolution^3 bits over the sampler's bounding box."""

    def __init__(self, bits: torch.Tensor, box_min, box_size, resolution: int):
        self.bits = bits
        self.box_min = [float(v) for v in box_min]
        self.box_size = [float(v) for v in box_size]
        self.resolution = int(resolution)

    @staticmethod
    def box_of(bounds: np.ndarray):
        """Axis-aligned box of a ``bounds`` transform (ray_sampler.py:101-104: the unit cube
        [-0.5, 0.5]^3 through the 4x4 matrix)."""
        corners = np.array([[x, y, z, 1.0] for x in (-0.5, 0.5) for y in (-0.5, 0.5)
                            for z in (-0.5, 0.5)], dtype=np.float64)
        world = (np.asarray(bounds, dtype=np.float64) @ corners.T).T[:, :3]
        lo, hi = world.min(axis=0), world.max(axis=0)
        return lo, hi - lo

    @staticmethod
    def cell_centres(bounds: np.ndarray, resolution: int, device) -> torch.Tensor:
        """(G^3,3) world positions of the cell centres, x fastest (the bit order)."""
        lo, size = OccupancyGrid.box_of(bounds)
        g = int(resolution)
        axis = [torch.arange(g, dtype=torch.float32, device=device).add_(0.5).mul_(float(size[d]) / g)
                .add_(float(lo[d])) for d in range(3)]
        zz, yy, xx = torch.meshgrid(axis[2], axis[1], axis[0], indexing="ij")
        return torch.stack([xx, yy, zz], dim=-1).reshape(-1, 3).contiguous()

    @classmethod
    def from_logits(cls, logits: torch.Tensor, bounds: np.ndarray, resolution: int,
                    sigma_threshold: float = 0.01, dilate: bool = True) -> "OccupancyGrid":
        """Grid from precomputed (G^3,4) raw outputs at ``cell_centres`` (any density source:
        a voxelised model, an analytic scene)."""
        lo, size = cls.box_of(bounds)
        bits = ops.occupancy_build(logits.contiguous(), int(resolution), float(sigma_threshold),
                                   bool(dilate))
        return cls(bits, lo, size, int(resolution))

    @classmethod
    def from_model(cls, model, bounds: np.ndarray, resolution: int = 128,
                   sigma_threshold: float = 0.01, dilate: bool = True,
                   batch_size: int = 1 << 21) -> "OccupancyGrid":
        """Evaluates the model's density at the cell centres (view direction +z for models
        that take one: sigma does not depend on it in nerf_model.py:118-119)."""
        device = next(model.parameters()).device
        lo, size = cls.box_of(bounds)
        g = int(resolution)
        axis = [torch.arange(g, dtype=torch.float32, device=device).add_(0.5).mul_(float(size[d]) / g)
                .add_(float(lo[d])) for d in range(3)]
        zz, yy, xx = torch.meshgrid(axis[2], axis[1], axis[0], indexing="ij")   # x fastest
        centres = torch.stack([xx, yy, zz], dim=-1).reshape(-1, 3).contiguous()
        logits = torch.empty((centres.shape[0], 4), dtype=torch.float32, device=device)
        was_training = model.training
        model.eval()
        with torch.no_grad():
            for start in range(0, centres.shape[0], batch_size):
                chunk = centres[start:start + batch_size]
                if getattr(model, "use_view", False):
                    view = torch.zeros_like(chunk)
                    view[:, 2] = 1.0
                    logits[start:start + batch_size] = model(chunk, view)
                else:
                    logits[start:start + batch_size] = model(chunk)
        model.train(was_training)
        bits = ops.occupancy_build(logits, g, float(sigma_threshold), bool(dilate))
        return cls(bits, lo, size, g)

    def fraction_occupied(self) -> float:
        """Share of cells marked occupied (diagnostic; one sync)."""
        cells = self.resolution ** 3
        words = self.bits.to(torch.int64) & 0xffffffff
        count = 0
        for shift in range(32):
            count += int(((words >> shift) & 1).sum().item())
        return count / cells

    def compact(self, positions: torch.Tensor, views: Optional[torch.Tensor]):
        return ops.occupancy_compact(positions, views, self.box_min, self.box_size,
                                     self.resolution, self.bits)
