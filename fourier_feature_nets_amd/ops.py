"""Thin tensor-level wrappers over the C ABI: argument checking, pointer plumbing, launch on
the tensors' own GPU and torch's current HIP stream of that GPU.  Every function requires
CUDA(HIP) float32 tensors; there is no CPU path."""

import ctypes
import math
from typing import Optional

import torch

from . import _lib
from ._lib import c_f, c_i, c_i64, c_p


class _DevPtr(ctypes.c_void_p):
    """A device pointer that remembers which GPU it points into (``device`` is None for a
    null pointer)."""
    device = None


def _dev(t: Optional[torch.Tensor], dtype=torch.float32, name="tensor"):
    if t is None:
        return _DevPtr(0)
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU (got %s); the HIP path has no CPU fallback"
                           % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    ptr = _DevPtr(t.data_ptr())
    ptr.device = t.device
    return ptr


def _call(name, *args):
    """Launches entry point ``name`` on the GPU its tensor arguments live on: that GPU is made
    current for the call (the C ABI launches on the calling thread's current device) and the
    kernel goes onto torch's current stream OF THAT GPU, which is appended as the trailing
    ``stream`` argument.  Tensors on different GPUs in one call are an error."""
    device = None
    for a in args:
        d = getattr(a, "device", None)
        if d is None:
            continue
        if device is None:
            device = d
        elif d != device:
            raise RuntimeError("%s: tensor arguments live on different devices (%s and %s)"
                               % (name, device, d))
    if device is None:
        raise RuntimeError("%s: no device tensor among the arguments" % name)
    stream = c_p(torch.cuda.current_stream(device).cuda_stream)
    if device.index == torch.cuda.current_device():
        return _lib.call(name, *args, stream)
    with torch.cuda.device(device):
        return _lib.call(name, *args, stream)


def _host3(values):
    arr = (ctypes.c_float * 3)(*[float(v) for v in values])
    return arr


# --------------------------------------------------------------------------------- rays
def raygen_nearfar(unproj: torch.Tensor, cam_pos: torch.Tensor, width: int, height: int,
                   box_lo, box_hi, points: Optional[torch.Tensor] = None):
    """K1.  unproj (C,4,4), cam_pos (C,3) on the GPU -> starts, directions, near_far, valid.
    ``points`` (W*H,2) float32 overrides the integer pixel grid."""
    cams = unproj.shape[0]
    total = cams * width * height
    dev = unproj.device
    starts = torch.empty((total, 3), dtype=torch.float32, device=dev)
    dirs = torch.empty((total, 3), dtype=torch.float32, device=dev)
    near_far = torch.empty((2, total), dtype=torch.float32, device=dev)
    valid = torch.empty((total,), dtype=torch.uint8, device=dev)
    _call("ffn_raygen_nearfar", _dev(unproj, name="unproj"), _dev(cam_pos, name="cam_pos"),
              _dev(points, name="points"), c_i(cams), c_i(width), c_i(height), _host3(box_lo), _host3(box_hi), _dev(starts),
              _dev(dirs), _dev(near_far), _dev(valid, torch.uint8))
    return starts, dirs, near_far, valid


def sample_t(near_far: torch.Tensor, ray_index: torch.Tensor, count: int, unit: torch.Tensor,
             noise: Optional[torch.Tensor], anneal: Optional[float],
             out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """K2a.  Returns (R, stride) t-values; the first `count` columns are filled."""
    rays = ray_index.shape[0]
    if out is None:
        out = torch.empty((rays, count), dtype=torch.float32, device=near_far.device)
    _call("ffn_sample_t", _dev(near_far), c_i64(near_far.shape[1]),
              _dev(ray_index, torch.int64, "ray_index"), c_i(rays), c_i(count), _dev(unit),
              _dev(noise), c_f(-1.0 if anneal is None else float(anneal)), _dev(out),
              c_i(out.shape[1]))
    return out


def materialise_samples(starts, directions, ray_index, t_values, want_views=True):
    """K2b.  positions (R,S,3) [and view_directions (R,S,3)]."""
    rays, count = t_values.shape
    pos = torch.empty((rays, count, 3), dtype=torch.float32, device=t_values.device)
    views = torch.empty_like(pos) if want_views else None
    _call("ffn_materialise_samples", _dev(starts), _dev(directions),
              _dev(ray_index, torch.int64), _dev(t_values), c_i(rays), c_i(count), _dev(pos),
              _dev(views))
    return pos, views


def sample_materialise(near_far, starts, directions, ray_index, count, unit, noise, anneal, want_views=True):
    """K2a + K2b in one launch: t (R,count), positions (R,count,3) [, view_directions]."""
    rays = ray_index.shape[0]
    t = torch.empty((rays, count), dtype=torch.float32, device=near_far.device)
    pos = torch.empty((rays, count, 3), dtype=torch.float32, device=near_far.device)
    views = torch.empty_like(pos) if want_views else None
    _call("ffn_sample_materialise", _dev(near_far), c_i64(near_far.shape[1]), _dev(starts), _dev(directions),
          _dev(ray_index, torch.int64, "ray_index"), c_i(rays), c_i(count), _dev(unit), _dev(noise),
          c_f(-1.0 if anneal is None else float(anneal)), _dev(t), _dev(pos), _dev(views))
    return t, pos, views


def cdf_build(t_probe: torch.Tensor, opacity: torch.Tensor) -> torch.Tensor:
    """K2c.  (P,n),(P,n) -> (P,n-1)."""
    rays, n = t_probe.shape
    cdf = torch.empty((rays, n - 1), dtype=torch.float32, device=t_probe.device)
    _call("ffn_cdf_build", _dev(t_probe), _dev(opacity), c_i64(rays), c_i(n), _dev(cdf))
    return cdf


def cdf_build_logits(t_probe: torch.Tensor, logits: torch.Tensor) -> torch.Tensor:
    """K2c on raw coarse-model outputs: (P,n), (P*n,4) -> (P,n-1); softplus inside."""
    rays, n = t_probe.shape
    cdf = torch.empty((rays, n - 1), dtype=torch.float32, device=t_probe.device)
    _call("ffn_cdf_build_logits", _dev(t_probe), _dev(logits), c_i64(rays), c_i(n), _dev(cdf))
    return cdf


def focus_sample_merge(near_far, cdfs, ray_index, u, unit_focus, t_io, n_focus, rows_local=False):
    """K2d.  In-place on t_io (R,S).  ``rows_local``: cdfs holds one row per batch ray."""
    rays, count = t_io.shape
    _call("ffn_focus_sample_merge_rows" if rows_local else "ffn_focus_sample_merge", _dev(near_far), c_i64(near_far.shape[1]), _dev(cdfs),
              _dev(ray_index, torch.int64), _dev(u), _dev(unit_focus), c_i(rays), c_i(count),
              c_i(n_focus), _dev(t_io))
    return t_io


def to_image(colors: torch.Tensor, pixel_index: torch.Tensor, width: int, height: int):
    """K8.  (n,3) colours + (n,) pixel ids -> (H,W,3) uint8 on the GPU."""
    img = torch.empty((height, width, 3), dtype=torch.uint8, device=colors.device)
    _call("ffn_to_image", _dev(colors), _dev(pixel_index, torch.int64),
              c_i64(colors.shape[0]), c_i(width), c_i(height), _dev(img, torch.uint8))
    return img


def ycrcb_to_rgb_u8(image: torch.Tensor) -> torch.Tensor:
    """K8b.  (H,W,3) uint8 YCrCb frame -> RGB, in place (OpenCV's 8-bit COLOR_YCrCb2RGB)."""
    _call("ffn_ycrcb_to_rgb_u8", _dev(image, torch.uint8, "image"), c_i64(image.numel() // 3))
    return image


# --------------------------------------------------------------------------------- encode
def fourier_encode(x: torch.Tensor, b: Optional[torch.Tensor], a: Optional[torch.Tensor],
                   scale: float, include_input: bool) -> torch.Tensor:
    """K3.  (N,3) -> (N, 2F[+3]) with the cos block first."""
    n = x.shape[0]
    freq = 0 if b is None else b.shape[1]
    width = 2 * freq + (3 if (include_input or freq == 0) else 0)
    out = torch.empty((n, width), dtype=torch.float32, device=x.device)
    _call("ffn_fourier_encode", _dev(x), c_i64(n), _dev(b), _dev(a), c_i(freq),
              c_f(scale), c_i(1 if include_input else 0), _dev(out))
    return out


# --------------------------------------------------------------------------------- composite
def composite_fwd(logits: torch.Tensor, t: torch.Tensor, include_depth: bool,
                  nan_flag: Optional[torch.Tensor] = None):
    """K5.  logits (R,S,4), t (R,S) -> color (R,3), alpha (R), depth (R)|None."""
    rays, count = t.shape
    dev = t.device
    color = torch.empty((rays, 3), dtype=torch.float32, device=dev)
    alpha = torch.empty((rays,), dtype=torch.float32, device=dev)
    depth = torch.empty((rays,), dtype=torch.float32, device=dev) if include_depth else None
    _call("ffn_composite_fwd", _dev(logits), _dev(t), c_i(rays), c_i(count), _dev(color),
              _dev(alpha), _dev(depth), _dev(nan_flag, torch.int32))
    return color, alpha, depth


def blend_weights(t: torch.Tensor, sigma: torch.Tensor) -> torch.Tensor:
    """K5w.  utils.calculate_blend_weights: (R,S),(R,S) -> (R,S)."""
    rays, count = t.shape
    out = torch.empty_like(t)
    _call("ffn_blend_weights", _dev(t), _dev(sigma), c_i(rays), c_i(count), _dev(out))
    return out


def blend_weights_bwd(t: torch.Tensor, sigma: torch.Tensor, d_weights: torch.Tensor,
                      want_dt: bool = False):
    """K5w backward.  Returns (d_sigma (R,S), d_t (R,S) | None)."""
    rays, count = t.shape
    d_sigma = torch.empty_like(t)
    d_t = torch.empty_like(t) if want_dt else None
    _call("ffn_blend_weights_bwd", _dev(t), _dev(sigma), _dev(d_weights), c_i(rays), c_i(count),
          _dev(d_sigma), _dev(d_t))
    return d_sigma, d_t


def composite_bwd(logits, t, d_color, d_alpha) -> torch.Tensor:
    """K5b.  d(logits) (R,S,4)."""
    rays, count = t.shape
    d_logits = torch.empty((rays, count, 4), dtype=torch.float32, device=t.device)
    _call("ffn_composite_bwd", _dev(logits), _dev(t), _dev(d_color), _dev(d_alpha),
              c_i(rays), c_i(count), _dev(d_logits))
    return d_logits


def mse_loss(color, alpha, gt_colors, gt_alphas, ray_index, color_scale, alpha_scale,
             want_grad=True, sums_out=None):
    """K6.  Returns (sums (2,), d_color, d_alpha); ``sums_out`` (2 floats) receives the sums
    in place of a fresh tensor."""
    rays = color.shape[0]
    dev = color.device
    sums = sums_out if sums_out is not None else torch.empty((2,), dtype=torch.float32, device=dev)
    scratch = torch.empty((2 * ((rays + 255) // 256),), dtype=torch.float32, device=dev)
    d_color = torch.empty_like(color) if want_grad else None
    d_alpha = torch.empty_like(alpha) if want_grad else None
    _call("ffn_mse_loss", _dev(color), _dev(alpha), _dev(gt_colors), _dev(gt_alphas),
              _dev(ray_index, torch.int64), c_i(rays), c_f(color_scale), c_f(alpha_scale),
              _dev(sums), _dev(d_color), _dev(d_alpha), _dev(scratch))
    return sums, d_color, d_alpha


def composite_train(logits, t, gt_colors, gt_alphas, ray_index, color_scale, alpha_scale,
                    nan_flag: Optional[torch.Tensor] = None):
    """K5t: K5 + K6 + K5b of one training batch in one launch.  logits (R,S,4), t (R,S) ->
    (d_logits (R,S,4), partials (blocks,2)): d_logits bit-identical to ``composite_fwd`` ->
    ``mse_loss`` -> ``composite_bwd``; the loss sums per workgroup go to ``loss_from_partials``."""
    rays, count = t.shape
    blocks = int(_lib.load().ffn_composite_train_blocks(c_i(rays)))
    d_logits = torch.empty((rays, count, 4), dtype=torch.float32, device=t.device)
    partials = torch.empty((blocks, 2), dtype=torch.float32, device=t.device)
    _call("ffn_composite_train", _dev(logits), _dev(t), c_i(rays), c_i(count), _dev(gt_colors),
          _dev(gt_alphas), _dev(ray_index, torch.int64), c_f(color_scale), c_f(alpha_scale),
          _dev(d_logits), _dev(partials), _dev(nan_flag, torch.int32))
    return d_logits, partials


def loss_from_partials(partials: torch.Tensor, rays: int, alpha_weight: float,
                       sums_out: Optional[torch.Tensor] = None, want_loss: bool = True):
    """Fixed-order sum of K5t's per-workgroup pairs: into ``sums_out`` (2 floats, for the
    data-parallel all-reduce) and / or the scalar loss (a fresh device scalar)."""
    loss = torch.empty((), dtype=torch.float32, device=partials.device) if want_loss else None
    _call("ffn_loss_from_partials", _dev(partials), c_i(partials.shape[0]), c_f(3.0 * rays),
          c_f(float(rays)), c_f(alpha_weight), _dev(sums_out), _dev(loss))
    return loss


def loss_value(sums: torch.Tensor, rays: int, alpha_weight: float) -> torch.Tensor:
    """sums[0] / (3 rays) + alpha_weight * sums[1] / rays as a fresh device scalar (one launch)."""
    out = torch.empty((), dtype=torch.float32, device=sums.device)
    _call("ffn_loss_value", _dev(sums), c_f(3.0 * rays), c_f(float(rays)), c_f(alpha_weight), _dev(out))
    return out


# --------------------------------------------------------------------------------- optimiser
def clip_adam(params, grads, exp_avg, exp_avg_sq, step: int, lr: float, weight_decay=0.0,
              clip_value=0.1, max_norm=0.1, beta1=0.9, beta2=0.999, eps=1e-8,
              scratch=None, norm_out=None):
    """K7 on flat fp32 buffers; `step` is the 1-based update count."""
    n = params.numel()
    if scratch is None:
        scratch = torch.empty(((n + 1023) // 1024,), dtype=torch.float32, device=params.device)
    step_size = lr / (1.0 - beta1 ** step)
    inv_sqrt_bc2 = 1.0 / math.sqrt(1.0 - beta2 ** step)
    _call("ffn_clip_adam", _dev(params), _dev(grads), _dev(exp_avg), _dev(exp_avg_sq),
              c_i64(n), c_f(clip_value), c_f(max_norm), c_f(step_size), c_f(inv_sqrt_bc2),
              c_f(beta1), c_f(beta2), c_f(eps), c_f(weight_decay), _dev(scratch),
              _dev(norm_out))


# --------------------------------------------------------------------------------- occupancy
def occupancy_build(logits: torch.Tensor, resolution: int, sigma_threshold: float,
                    dilate: bool) -> torch.Tensor:
    """K9a/b.  logits (G^3,4) at the cell centres -> bit mask (ceil(G^3/32),) int32."""
    words = (resolution ** 3 + 31) // 32
    bits = torch.empty((words,), dtype=torch.int32, device=logits.device)
    scratch = torch.empty_like(bits) if dilate else None
    _call("ffn_occupancy_build", _dev(logits), c_i(resolution), c_f(sigma_threshold),
              c_i(1 if dilate else 0), _dev(scratch, torch.int32), _dev(bits, torch.int32))
    return bits


def occupancy_compact(positions: torch.Tensor, views: Optional[torch.Tensor], box_min, box_size,
                      resolution: int, bits: torch.Tensor):
    """K9c-e.  (N,3) samples -> packed positions / views of the samples in occupied cells and
    their int32 source index.  One device-to-host sync (the packed count sizes the outputs)."""
    n = positions.shape[0]
    dev = positions.device
    blocks = (n + 255) // 256
    offsets = torch.empty((blocks,), dtype=torch.int32, device=dev)
    total = torch.empty((1,), dtype=torch.int64, device=dev)
    lo, size = _host3(box_min), _host3(box_size)
    _call("ffn_occupancy_count", _dev(positions), c_i64(n), lo, size, c_i(resolution),
              _dev(bits, torch.int32), _dev(offsets, torch.int32), _dev(total, torch.int64))
    m = int(total.item())
    out_pos = torch.empty((m, 3), dtype=torch.float32, device=dev)
    out_view = torch.empty((m, 3), dtype=torch.float32, device=dev) if views is not None else None
    index = torch.empty((m,), dtype=torch.int32, device=dev)
    if m > 0:
        _call("ffn_occupancy_compact", _dev(positions), _dev(views), c_i64(n), lo, size,
                  c_i(resolution), _dev(bits, torch.int32), _dev(offsets, torch.int32),
                  _dev(out_pos), _dev(out_view), _dev(index, torch.int32))
    return out_pos, out_view, index


def scatter_logits(packed: torch.Tensor, index: torch.Tensor, n: int,
                   empty_sigma_logit: float = -100.0) -> torch.Tensor:
    """K9f.  (M,4) logits of the evaluated samples -> (N,4), the rest (0,0,0,empty_sigma_logit)."""
    out = torch.empty((n, 4), dtype=torch.float32, device=index.device)
    _call("ffn_scatter_logits", _dev(packed), _dev(index, torch.int32), c_i64(packed.shape[0]),
              c_i64(n), c_f(empty_sigma_logit), _dev(out))
    return out


def gather_logits(full: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """K9g.  (N,4) rows at ``index`` (int32) -> (M,4)."""
    m = index.shape[0]
    out = torch.empty((m, 4), dtype=torch.float32, device=full.device)
    if m > 0:
        _call("ffn_gather_logits", _dev(full), _dev(index, torch.int32), c_i64(m), _dev(out))
    return out


# --------------------------------------------------------------------------------- voxels
def voxels_forward(volume: torch.Tensor, bias: torch.Tensor, positions: torch.Tensor, side: int,
                   scale: float) -> torch.Tensor:
    """K10.  volume (4,S,S,S), bias (4), positions (N,3) -> logits (N,4)."""
    n = positions.shape[0]
    out = torch.empty((n, 4), dtype=torch.float32, device=positions.device)
    _call("ffn_voxels_forward", _dev(volume), _dev(bias), _dev(positions, name="positions"),
              c_i64(n), c_i(side), c_f(scale), _dev(out))
    return out
