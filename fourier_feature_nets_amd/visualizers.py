"""Training-time visualizers behind the names the reference drivers use
(``train_nerf.py:109-130``, ``train_tiny_nerf.py``): hooks with a
``visualize(step, render, act_render)`` method that ``Raycaster.fit`` calls every step.

They are callers of the hot path, not part of it: each renders one camera through the render
function ``fit`` hands over (``Raycaster.batched_render``, i.e. the HIP kernels) and writes a PNG
with PIL (the reference uses OpenCV, which this image does not have).  File names and layout
follow ``visualizers.py:33-153`` of the reference: ``<results>/<label>/s{step:07}_c{cam:03}.png``
with the 2x2 grid predicted | depth over actual | error, and ``<results>/video/frame_{i:05d}.png``.
"""

import os

import numpy as np

from .cameras import Resolution, orbit
from .sampler import RaySampler


def _save_png(path: str, image: np.ndarray):
    from PIL import Image
    Image.fromarray(image).save(path)


class Visualizer:
    """Base class: called with the optimisation step and the two render functions."""

    def visualize(self, step: int, render, act_render):
        raise NotImplementedError


class EvaluationVisualizer(Visualizer):
    """Prediction, depth, ground truth and error of one camera as a 2x2 grid, every
    ``interval`` steps, walking through the dataset's cameras."""

    def __init__(self, results_dir: str, dataset, interval: int, max_depth=10):
        self._output_dir = os.path.join(results_dir, dataset.label)
        os.makedirs(self._output_dir, exist_ok=True)
        self._dataset = dataset
        self._interval = interval
        self._index = 0
        self._max_depth = max_depth

    def visualize(self, step: int, render, _):
        if step % self._interval != 0:
            return
        ds = self._dataset
        camera = self._index % ds.num_cameras
        samples = ds.rays_for_camera(camera)
        truth = ds.render(samples).numpy()
        pred = render(samples, True)
        error = np.square(truth.color - pred.color).sum(-1)
        if truth.alpha is not None:
            error = (3 * error + np.square(truth.alpha - pred.alpha)) / 4
        width, height = ds.cameras[camera].resolution
        predicted = ds.to_image(camera, np.clip(pred.color, 0, 1))
        shown = truth.color if truth.alpha is None else truth.color * truth.alpha[..., np.newaxis]
        actual = ds.to_image(camera, shown)
        depth = ds.to_image(camera, np.clip(pred.depth, 0, self._max_depth) / self._max_depth)
        error = np.sqrt(error)
        peak = float(error.max()) if error.size else 0.0
        error = ds.to_image(camera, error / peak if peak > 0 else error)
        grid = np.zeros((height * 2, width * 2, 3), np.uint8)
        grid[:height, :width] = predicted
        grid[height:, :width] = actual
        grid[:height, width:] = depth
        grid[height:, width:] = error
        _save_png(os.path.join(self._output_dir, "s{:07}_c{:03}.png".format(step, camera)), grid)
        self._index += 1


class OrbitVideoVisualizer(Visualizer):
    """One frame of an orbit around the volume every ``num_steps // num_frames`` steps."""

    def __init__(self, results_dir: str, num_steps: int, resolution: Resolution, num_frames: int,
                 num_samples: int, color_space: str, device=None):
        self._output_dir = os.path.join(results_dir, "video")
        os.makedirs(self._output_dir, exist_ok=True)
        cameras = orbit(np.array([0, 1, 0]), np.array([0, 0, -1]), num_frames, 40,
                        resolution.square(), 4)
        bounds = np.eye(4, dtype=np.float32) * 2
        kwargs = {} if device is None else {"device": device}
        self._sampler = RaySampler(bounds, cameras, num_samples, **kwargs)
        self._interval = max(1, num_steps // num_frames)
        self._index = 0
        self._color_space = color_space

    def visualize(self, step: int, render, _):
        if step % self._interval != 0:
            return
        camera = self._index % self._sampler.num_cameras
        pred = render(self._sampler.rays_for_camera(camera), False)
        image = self._sampler.to_image(camera, pred.color, self._color_space)
        _save_png(os.path.join(self._output_dir, "frame_{:05d}.png".format(self._index)), image)
        self._index += 1


class ActivationVisualizer(Visualizer):
    """Per-layer activation videos of the lecture notes: needs ``render_activations``, which is
    outside the HIP hot path."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("ActivationVisualizer needs Raycaster.render_activations, a "
                                  "lecture visualisation outside the HIP hot path")
