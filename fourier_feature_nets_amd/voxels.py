"""Dense voxel radiance field ("voxels" checkpoints of the reference, voxels_model.py:9-56),
here only ever an *opacity model* for the focus sampler: a (1,4,S,S,S) volume of raw
[r,g,b,sigma] logits over the cube [-scale, scale]^3, looked up trilinearly by the HIP kernel K10
(``csrc/occupancy.hip``).  State-dict keys (``voxels``, ``bias``), ``params`` and the ``save``
format are the reference's, so its checkpoints load; optimising the volume itself (the
reference's voxel training script) is outside the hot path and raises."""

import torch
import torch.nn as nn

from . import ops


class Voxels(nn.Module):
    use_view = False

    def __init__(self, side: int, scale: float):
        super().__init__()
        self.params = dict(side=side, scale=scale)
        self.scale = scale
        self.voxels = nn.Parameter(torch.zeros((1, 4, side, side, side), dtype=torch.float32))
        # the reference starts from (almost) black and sigma logit -2
        start = torch.full((4,), float(torch.logit(torch.tensor(1e-5))), dtype=torch.float32)
        start[3] = -2.0
        self.bias = nn.Parameter(start.reshape(1, 4))

    def forward(self, positions: torch.Tensor) -> torch.Tensor:
        """(N,3) world positions -> (N,4) raw logits."""
        if torch.is_grad_enabled() and (self.voxels.requires_grad or self.bias.requires_grad):
            raise NotImplementedError("Voxels is an inference-only opacity model here (wrap the "
                                      "call in torch.no_grad()); training the volume is outside "
                                      "the HIP hot path")
        side = self.voxels.shape[-1]
        out = ops.voxels_forward(self.voxels.detach().reshape(4, side, side, side).contiguous(),
                                 self.bias.detach().reshape(4).contiguous(),
                                 positions.reshape(-1, 3).contiguous(), side, float(self.scale))
        return out

    def save(self, path: str):
        blob = self.state_dict()
        blob["type"] = "voxels"
        blob["params"] = self.params
        torch.save(blob, path)
