"""Small shared pieces of the reference's ``utils`` surface: blend weights, linspace, the
learning-rate schedule, checkpoint loading, result tuples and a quiet progress bar."""

import os
import sys
import time
from typing import NamedTuple, Optional

import torch

from . import ops


class _BlendWeights(torch.autograd.Function):
    """w = alpha * exclusive-cumprod(tau) (kernel K5w) with its closed-form backward (K5w-b)."""

    @staticmethod
    def forward(ctx, t_values, opacity):
        t_values = t_values.contiguous()
        opacity = opacity.contiguous()
        ctx.save_for_backward(t_values, opacity)
        return ops.blend_weights(t_values, opacity)

    @staticmethod
    def backward(ctx, d_weights):
        t_values, opacity = ctx.saved_tensors
        d_sigma, d_t = ops.blend_weights_bwd(t_values, opacity, d_weights.contiguous(),
                                             want_dt=ctx.needs_input_grad[0])
        return d_t, d_sigma


def calculate_blend_weights(t_values: torch.Tensor, opacity: torch.Tensor) -> torch.Tensor:
    """Front-to-back alpha-compositing weights, differentiable w.r.t. the opacities (and the
    t-values) like the reference's (utils.py:72-97)."""
    return _BlendWeights.apply(t_values, opacity)


def linspace(start: torch.Tensor, stop: torch.Tensor, num_samples: int) -> torch.Tensor:
    """(D,) start/stop -> (D, num_samples), both ends included (utils.py:179-194).
    Runs kernel K2a with identity indexing, so the arithmetic is the sampler's own."""
    rows = start.shape[0]
    near_far = torch.stack([start, stop]).contiguous()
    index = torch.arange(rows, dtype=torch.int64, device=start.device)
    unit = torch.linspace(0, 1, num_samples).to(start.device)
    return ops.sample_t(near_far, index, num_samples, unit, None, None)


def exponential_lr_decay(optim, initial_learning_rate: float, step: int, decay_rate: float,
                         decay_steps: float):
    """lr = lr0 * rate ** (step / decay_steps) written into every param group
    (utils.py:422-445)."""
    lr = learning_rate_at(initial_learning_rate, step, decay_rate, decay_steps)
    for group in optim.param_groups:
        group["lr"] = lr


def learning_rate_at(initial_learning_rate: float, step: int, decay_rate: float,
                     decay_steps: float) -> float:
    return initial_learning_rate * decay_rate ** (step / decay_steps)


def rgb_to_ycrcb_u8(rgb):
    """(…,3) uint8 RGB -> uint8 YCrCb as cv2.cvtColor(image, cv2.COLOR_RGB2YCrCb) does for 8-bit
    images (image_dataset.py:114-115): OpenCV's 14-bit fixed-point path -- Y = round(0.299 R +
    0.587 G + 0.114 B), Cr = round(0.713 (R - Y)) + 128, Cb = round(0.564 (B - Y)) + 128 with the
    coefficients scaled by 2^14 and rounded (4899, 9617, 1868, 11682, 9241), "round" = add 2^13
    and shift right by 14 (arithmetic), saturating cast.  One-time host work at dataset
    construction, like the reference's.  cv2 is not in this image: restated from OpenCV's
    documented constants, parity unpinned."""
    import numpy as np
    v = np.asarray(rgb).astype(np.int32)
    r, g, b = v[..., 0], v[..., 1], v[..., 2]
    half = 1 << 13
    y = (r * 4899 + g * 9617 + b * 1868 + half) >> 14
    cr = ((r - y) * 11682 + (128 << 14) + half) >> 14
    cb = ((b - y) * 9241 + (128 << 14) + half) >> 14
    return np.clip(np.stack([y, cr, cb], -1), 0, 255).astype(np.uint8)


COLOR_SPACES = ("RGB", "YCrCb")


def check_color_space(color_space: str) -> str:
    if color_space not in COLOR_SPACES:
        raise NotImplementedError("Unsupported color space: {}".format(color_space))
    return color_space


class RenderResult(NamedTuple("RenderResult", [("color", torch.Tensor), ("alpha", torch.Tensor),
                                               ("depth", torch.Tensor)])):
    """Per-ray colour, alpha and (optionally) depth."""

    @property
    def device(self) -> torch.device:
        return self.color.device

    def to(self, *args) -> "RenderResult":
        return RenderResult(*[None if x is None else x.to(*args) for x in self])

    def numpy(self) -> "RenderResult":
        return RenderResult(*[None if x is None else x.detach().cpu().numpy() for x in self])


class ETABar:
    """Minimal stand-in for the reference's progress bar (utils.py:36-69): same calls, one
    line of text per update when attached to a terminal, silent otherwise."""

    def __init__(self, message: str, max: int = 100):
        self.message = message
        self.max = max
        self.index = 0
        self._start = time.time()
        self._tty = sys.stdout.isatty()
        self._info = ""

    def next(self, n: int = 1):
        self.index += n
        if self._tty:
            elapsed = time.time() - self._start
            eta = elapsed / max(self.index, 1) * (self.max - self.index)
            sys.stdout.write("\r%s %d/%d - %ds %s" % (self.message, self.index, self.max, eta,
                                                      self._info))
            sys.stdout.flush()

    def info(self, text: str):
        self._info = text

    def finish(self):
        if self._tty:
            sys.stdout.write("\n")


def load_model(path: str) -> Optional[torch.nn.Module]:
    """Loads a checkpoint written by ``model.save`` (ours or the reference's; same format:
    state dict + "type" + "params", utils.py:448-503) and returns the model in eval mode,
    or ``None`` (with a message) when the file does not exist.  No download is attempted."""
    from .models import FourierFeatureMLP, NeRF
    from .voxels import Voxels
    if not os.path.exists(path):
        alt = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "models", path))
        if not os.path.exists(alt):
            print("Unable to find model", path)
            return None
        path = alt
    blob = torch.load(path, map_location="cpu")
    kind = blob.pop("type")
    params = blob.pop("params")
    if kind == "fourier":
        for key in ("a_values", "b_values"):
            if params[key] is not None:
                params[key] = torch.FloatTensor(params[key])
        model = FourierFeatureMLP(**params)
    elif kind == "nerf":
        model = NeRF(**params)
    elif kind == "voxels":
        model = Voxels(**params)
    else:
        print("Unrecognized model type:", kind)
        return None
    model.load_state_dict(blob)
    model.eval()
    return model
