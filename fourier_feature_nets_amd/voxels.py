"""Dense voxel radiance field, usable as the ``opacity_model`` of the focus sampler.

Out of scope for hand-written kernels: the trilinear lookup is
the stock ``grid_sample`` of PyTorch-ROCm.  Kept so that checkpoints of type "voxels"
(reference voxels_model.py:9-56) load and can drive opacity-guided sampling."""

import torch
import torch.nn as nn
import torch.nn.functional as F


class Voxels(nn.Module):
    def __init__(self, side: int, scale: float):
        nn.Module.__init__(self)
        self.params = {"side": side, "scale": scale}
        self.voxels = nn.Parameter(torch.zeros((1, 4, side, side, side), dtype=torch.float32))
        bias = torch.zeros(4, dtype=torch.float32)
        bias[:3] = torch.logit(torch.FloatTensor([1e-5, 1e-5, 1e-5]))
        bias[3] = -2
        self.bias = nn.Parameter(bias.unsqueeze(0))
        self.scale = scale
        self.use_view = False

    def forward(self, positions: torch.Tensor) -> torch.Tensor:
        grid = (positions / self.scale).reshape(1, -1, 1, 1, 3)
        out = F.grid_sample(self.voxels, grid, padding_mode="border", align_corners=False)
        out = out.transpose(1, 2).reshape(-1, 4) + self.bias
        assert not out.isnan().any()
        return out

    def save(self, path: str):
        blob = self.state_dict()
        blob["type"] = "voxels"
        blob["params"] = self.params
        torch.save(blob, path)
