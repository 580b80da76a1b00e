"""CPU oracle for the NeRF volume-rendering hot path.  TEST INFRASTRUCTURE ONLY.

This module is a from-scratch CPU restatement (numpy + torch-CPU float32) of the
arithmetic on the hot path of matajoh/fourier_feature_nets.  It exists so that the
HIP kernels in ``fourier_feature_nets_amd/csrc`` can be checked against something
that runs anywhere.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package never does
(it raises when the HIP library or a GPU is missing).

Parity status: PINNED.  The reference has no tests of its own, so every function
below is checked (``tests/test_oracle_golden.py``) against fixtures under
``tests/golden/`` that were produced by importing the reference itself in the build
container (``tests/golden/make_goldens.py``; torch 2.10.0 CPU, numpy 2.2.6).

Each function cites the reference lines (paths relative to the reference root) it
restates.  Layout conventions: rays are numbered ``cam*W*H + y*W + x``; every float
is float32; ``near_far`` is stored (2, num_rays) like the reference.
"""

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- #
#  a1 / a2: ray generation and slab test
# --------------------------------------------------------------------------- #

def unprojection(intrinsics: np.ndarray, extrinsics: np.ndarray) -> np.ndarray:
    """4x4 pixel->world matrix.  Restates camera_info.py:66-70 (unproject).

    projection = [[K,0],[0,1]] @ inv(E); unprojection = inv(projection), both in
    float32 through numpy's LAPACK path exactly as the reference does.
    """
    proj = np.eye(4, dtype=np.float32)
    proj[:3, :3] = intrinsics[:3, :3]
    proj = proj @ np.linalg.inv(extrinsics)
    return np.linalg.inv(proj)


def pixel_grid(width: int, height: int) -> np.ndarray:
    """Integer pixel coordinates, x fastest.  Restates ray_sampler.py:133-136."""
    xs, ys = np.meshgrid(np.arange(width), np.arange(height))
    return np.stack([xs, ys], -1).reshape(-1, 2)


def raycast(intrinsics: np.ndarray, extrinsics: np.ndarray,
            points: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Ray origins and unit directions.  Restates camera_info.py:99-109.

    world = U @ [x, y, 1, 1]^T with integer pixel coordinates (no +0.5);
    direction = normalise(world[:3] - camera_position); origin = camera position
    broadcast through ``+ 0 * dir`` (kept: it turns inf/nan directions into nan
    origins the same way the reference does).
    """
    pts = points.astype(np.float32).reshape(-1, 2)
    unproj = unprojection(intrinsics, extrinsics)
    homog = np.concatenate([pts, np.ones((pts.shape[0], 2), np.float32)], -1)
    world = (unproj @ homog.T).T
    cam = extrinsics[:3, 3].reshape(1, 3)
    delta = world[:, :3] - cam
    direction = delta / np.linalg.norm(delta, axis=-1, keepdims=True)
    return cam + 0 * direction, direction


def aabb_from_bounds(bounds: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Axis-aligned box corners.  Restates ray_sampler.py:101-104."""
    lo = bounds @ np.array([-0.5, -0.5, -0.5, 1], np.float32)
    hi = bounds @ np.array([0.5, 0.5, 0.5, 1], np.float32)
    return lo[np.newaxis, :3], hi[np.newaxis, :3]


def near_far(starts: np.ndarray, directions: np.ndarray, box_lo: np.ndarray,
             box_hi: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Slab test.  Restates ray_sampler.py:202-232 (_near_far).

    Returns (near_far (2,P) float32, valid (P,) bool).  near is clamped to >= 0.1
    for valid rays only; invalid rays keep their near >= far (or NaN) values.
    """
    with np.errstate(divide="ignore", invalid="ignore"):
        t_lo = (box_lo - starts) / directions
        t_hi = (box_hi - starts) / directions
    near = np.where(t_lo < t_hi, t_lo, t_hi).max(-1)
    far = np.where(t_lo > t_hi, t_lo, t_hi).min(-1)
    valid = near < far
    near[valid] = np.maximum(0.1, near[valid])
    return np.stack([near, far]), valid


def sampler_state(bounds: np.ndarray, intrinsics: Sequence[np.ndarray],
                  extrinsics: Sequence[np.ndarray], width: int,
                  height: int) -> Dict[str, object]:
    """All per-ray sampler state.  Restates ray_sampler.py:100-170 (no focus part)."""
    lo, hi = aabb_from_bounds(bounds)
    pts = pixel_grid(width, height)
    starts, dirs, nfs, invalid = [], [], [], []
    base = 0
    for k_mat, e_mat in zip(intrinsics, extrinsics):
        s, d = raycast(k_mat, e_mat, pts)
        nf, ok = near_far(s, d, lo, hi)
        starts.append(torch.from_numpy(s.astype(np.float32)))
        dirs.append(torch.from_numpy(d.astype(np.float32)))
        nfs.append(torch.from_numpy(nf.astype(np.float32)))
        invalid.append(np.nonzero(~ok)[0].astype(np.int64) + base)
        base += len(pts)
    return {
        "starts": torch.cat(starts),
        "directions": torch.cat(dirs),
        "near_far": torch.cat(nfs, -1),
        "invalid": np.concatenate(invalid) if invalid else np.zeros(0, np.int64),
        "rays_per_camera": width * height,
        "num_rays": base,
    }


# --------------------------------------------------------------------------- #
#  a4 / a5: t sampling
# --------------------------------------------------------------------------- #

def linspace_rows(start: torch.Tensor, stop: torch.Tensor, count: int) -> torch.Tensor:
    """Per-row linspace incl. both ends.  Restates utils.py:179-194.

    A multiply then an add (two roundings, no fused multiply-add).
    """
    unit = torch.linspace(0, 1, count)
    return start.unsqueeze(-1) + unit.unsqueeze(0) * (stop - start).unsqueeze(-1)


def anneal_range(near: torch.Tensor, far: torch.Tensor, step: Optional[int],
                 anneal_start: float, num_anneal_steps: int):
    """Pulls [near, far] toward the midpoint early in training.

    Restates ray_sampler.py:373-378.
    """
    if step is not None and step < num_anneal_steps:
        factor = min(max(step / num_anneal_steps, anneal_start), 1)
        mid = (near + far) * 0.5
        near = mid + (near - mid) * factor
        far = mid + (far - mid) * factor
    return near, far


def uniform_t(near: torch.Tensor, far: torch.Tensor, count: int,
              noise: Optional[torch.Tensor]) -> torch.Tensor:
    """Evenly spaced (optionally jittered) t.  Restates ray_sampler.py:380-386.

    ``noise`` is the (R, count) block the reference draws with ``torch.rand``;
    passing it in keeps the oracle deterministic and lets the device path be
    compared bit for bit.
    """
    t = linspace_rows(near, far, count)
    if noise is not None:
        scale = (far - near) / count
        t = t + noise * scale.unsqueeze(-1)
    return t


def determine_cdf(t_values: torch.Tensor, opacity: torch.Tensor) -> torch.Tensor:
    """Per-ray CDF of interior blend weights.  Restates ray_sampler.py:59-67."""
    w = blend_weights(t_values, opacity)[:, 1:-1] + 1e-5
    cdf = w.cumsum(-1)
    cdf = cdf / cdf[:, -1:]
    return torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)


def focus_t(near: torch.Tensor, far: torch.Tensor, cdf: torch.Tensor,
            u: torch.Tensor) -> torch.Tensor:
    """Inverse-transform samples from the CDF.  Restates ray_sampler.py:301-357.

    Bin locations are the mid-points of linspace(near, far, n); ``u`` is the
    (R, n) uniform block (``torch.rand`` when stratified, ``linspace(0,1,n)``
    repeated otherwise).
    """
    n = u.shape[1]
    grid = linspace_rows(near, far, n)
    mids = 0.5 * (grid[..., :-1] + grid[..., 1:])
    k = torch.searchsorted(cdf, u, right=True)
    lo = torch.clamp(k - 1, min=0)
    hi = torch.clamp(k, max=cdf.shape[-1] - 1)
    c_lo, c_hi = torch.gather(cdf, 1, lo), torch.gather(cdf, 1, hi)
    t_lo, t_hi = torch.gather(mids, 1, lo), torch.gather(mids, 1, hi)
    denom = c_hi - c_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    frac = (u - c_lo) / denom
    return t_lo + frac * (t_hi - t_lo)


def sample(state: Dict[str, object], idx, step: Optional[int], num_samples: int,
           anneal_start: float = 0.5, num_anneal_steps: int = 0,
           noise: Optional[torch.Tensor] = None, cdfs: Optional[torch.Tensor] = None,
           focus_u: Optional[torch.Tensor] = None):
    """RaySampler.sample.  Restates ray_sampler.py:359-403.

    Returns (positions (R,S,3), view_directions (R,S,3), t_values (R,S),
    rays (R,) int64).  ``noise`` != None means stratified.  With ``cdfs`` the
    first S//2 samples are uniform, the remaining S - S//2 come from the CDF
    (``focus_u``), and the union is sorted.
    """
    idx_t = torch.as_tensor(np.asarray(idx), dtype=torch.long)
    starts = state["starts"][idx_t]
    dirs = state["directions"][idx_t]
    near, far = state["near_far"][:, idx_t]
    n_uniform = num_samples // 2 if cdfs is not None else num_samples
    a_near, a_far = anneal_range(near, far, step, anneal_start, num_anneal_steps)
    t = uniform_t(a_near, a_far, n_uniform, noise)
    if cdfs is not None:
        extra = focus_t(near, far, cdfs[idx_t], focus_u)
        t, _ = torch.cat([t, extra], -1).sort(-1)
    rays = len(idx_t)
    dirs_rs = dirs.reshape(rays, 1, 3).repeat(1, num_samples, 1)
    positions = starts.reshape(rays, 1, 3) + t.unsqueeze(-1) * dirs_rs
    return positions, dirs_rs, t, idx_t


def voxels_forward(voxels: torch.Tensor, bias: torch.Tensor, scale: float,
                   positions: torch.Tensor) -> torch.Tensor:
    """Raw (N,4) logits of a dense voxel radiance field: trilinear lookup of ``voxels``
    (1,4,S,S,S) at ``positions / scale`` (border padding, cell-centre sample grid) plus
    ``bias`` (1,4).  Restates voxels_model.py:35-45."""
    grid = (positions.reshape(1, -1, 1, 1, 3) / scale)
    out = F.grid_sample(voxels, grid, padding_mode="border", align_corners=False)
    return out.transpose(1, 2).reshape(-1, 4) + bias


def opacity_cdfs(state: Dict[str, object], num_samples: int, opacity_fn, batch: int = 4096) -> torch.Tensor:
    """The per-ray CDF table an opacity-guided sampler builds at construction: probe points
    t = linspace(near, far, S_f) with S_f = S - S // 2, sigma = softplus(last output of the
    opacity model), blend-weight CDF.  Restates ray_sampler.py:59-67,148-166,234-269
    (``opacity_fn``: (N,3) positions -> (N,C) raw outputs)."""
    n_focus = num_samples - num_samples // 2
    near, far = state["near_far"]
    t_probe = linspace_rows(near, far, n_focus)
    rows = []
    with torch.no_grad():
        for lo in range(0, t_probe.shape[0], batch):
            t = t_probe[lo:lo + batch]
            pos = state["starts"][lo:lo + batch].unsqueeze(1) + t.unsqueeze(2) * state["directions"][lo:lo + batch].unsqueeze(1)
            sigma = F.softplus(opacity_fn(pos.reshape(-1, 3))[:, -1]).reshape(t.shape)
            rows.append(determine_cdf(t, sigma))
    return torch.cat(rows)


# --------------------------------------------------------------------------- #
#  a8 / a9: encodings and MLPs
# --------------------------------------------------------------------------- #

def axis_frequency_matrix(max_log_scale: float, num_freq: int,
                          num_inputs: int = 3) -> torch.Tensor:
    """(num_inputs, num_inputs*num_freq) block matrix, columns interleaved per
    frequency [x f0, y f0, z f0, x f1, ...].

    Restates nerf_model.py:76-84 and fourier_feature_models.py:157-166.
    """
    freqs = 2. ** torch.linspace(0, max_log_scale, num_freq)
    mat = torch.eye(num_inputs) * freqs.reshape(-1, 1, 1)
    return mat.reshape(-1, num_inputs).transpose(0, 1)


def positional_b_values(max_log_scale: float, embedding_size: int,
                        num_inputs: int = 3) -> torch.Tensor:
    """Restates fourier_feature_models.py:157-166 (embedding_size // num_inputs)."""
    return axis_frequency_matrix(max_log_scale, embedding_size // num_inputs, num_inputs)


def fourier_features(x: torch.Tensor, a_values: Optional[torch.Tensor],
                     b_values: Optional[torch.Tensor]) -> torch.Tensor:
    """gamma(x) = [a cos(pi x B), a sin(pi x B)], cos first.

    Restates fourier_feature_models.py:59-68.
    """
    if b_values is None:
        return x
    e = (math.pi * x) @ b_values
    return torch.cat([a_values * e.cos(), a_values * e.sin()], -1)


def fourier_mlp_forward(x: torch.Tensor, a_values, b_values,
                        weights: Sequence[torch.Tensor],
                        biases: Sequence[torch.Tensor]) -> torch.Tensor:
    """FourierFeatureMLP.forward.  Restates fourier_feature_models.py:57-78."""
    h = fourier_features(x, a_values, b_values)
    for w, b in zip(weights[:-1], biases[:-1]):
        h = torch.relu(F.linear(h, w, b))
    return F.linear(h, weights[-1], biases[-1])


def fourier_mlp_last_hidden(x: torch.Tensor, a_values, b_values,
                            weights: Sequence[torch.Tensor],
                            biases: Sequence[torch.Tensor]) -> torch.Tensor:
    """What `FourierFeatureMLP.keep_activations` records: the output of the last hidden layer.
    Restates fourier_feature_models.py:70-75."""
    h = fourier_features(x, a_values, b_values)
    for w, b in zip(weights[:-1], biases[:-1]):
        h = torch.relu(F.linear(h, w, b))
    return h


def nerf_encode(x: torch.Tensor, enc: torch.Tensor, include_inputs: bool) -> torch.Tensor:
    """[cos(xB), sin(xB), x] with no pi factor.  Restates nerf_model.py:97-102."""
    e = x @ enc
    parts = [e.cos(), e.sin()]
    if include_inputs:
        parts.append(x)
    return torch.cat(parts, -1)


def nerf_forward(position: torch.Tensor, view: torch.Tensor,
                 p: Dict[str, torch.Tensor], skips: Sequence[int],
                 include_inputs: bool) -> torch.Tensor:
    """NeRF.forward.  Restates nerf_model.py:86-124.

    ``p`` uses the reference state-dict names (layers.N.weight, opacity_out.*,
    bottleneck.*, hidden_view.*, color_out.*, pos_encoding, view_encoding).
    """
    enc_pos = nerf_encode(position, p["pos_encoding"], include_inputs)
    enc_view = nerf_encode(view, p["view_encoding"], include_inputs)
    h = enc_pos
    i = 0
    while "layers.%d.weight" % i in p:
        if i in set(skips):
            h = torch.cat([h, enc_pos], -1)
        h = torch.relu(F.linear(h, p["layers.%d.weight" % i], p["layers.%d.bias" % i]))
        i += 1
    sigma = F.linear(h, p["opacity_out.weight"], p["opacity_out.bias"])
    neck = F.linear(h, p["bottleneck.weight"], p["bottleneck.bias"])
    hv = torch.relu(F.linear(torch.cat([neck, enc_view], -1),
                             p["hidden_view.weight"], p["hidden_view.bias"]))
    rgb = F.linear(hv, p["color_out.weight"], p["color_out.bias"])
    return torch.cat([rgb, sigma], -1)


# --------------------------------------------------------------------------- #
#  a10 / a11 / a12: compositing and loss
# --------------------------------------------------------------------------- #

def blend_weights(t_values: torch.Tensor, opacity: torch.Tensor) -> torch.Tensor:
    """Front-to-back alpha-compositing weights.  Restates utils.py:72-97.

    delta_last = 1e10; alpha = 1 - exp(-(sigma*delta)); tau = min(1, 1-alpha+1e-10);
    T = exclusive cumprod(tau); w = alpha * T.
    """
    count = t_values.shape[1]
    delta = t_values[:, 1:] - t_values[:, :-1]
    delta = torch.cat([delta, torch.full_like(delta[:, :1], 1e10)], -1)
    alpha = 1 - torch.exp(-(opacity * delta))
    tau = torch.minimum(torch.ones_like(alpha), 1 - alpha + 1e-10)
    tau = torch.cat([torch.ones_like(tau[:, :1]), tau[:, :count - 1]], -1)
    return alpha * torch.cumprod(tau, -1)


def render(logits: torch.Tensor, t_values: torch.Tensor, include_depth: bool = True):
    """Raycaster.render after the model call.  Restates ray_caster.py:66-93.

    logits: (R,S,4) raw [r,g,b,sigma].  Colour sums all S weights; alpha and depth
    ignore the last sample; a ray with alpha < 0.1 reports t[:, -1] as depth.
    """
    rgb = torch.sigmoid(logits[..., :3])
    sigma = F.softplus(logits[..., 3])
    assert not rgb.isnan().any()
    assert not sigma.isnan().any()
    w = blend_weights(t_values, sigma)
    color = (w.unsqueeze(-1) * rgb).sum(-2)
    w_in = w[:, :-1]
    alpha = w_in.sum(-1)
    depth = None
    if include_depth:
        pick = w_in.argmax(-1)
        pick[alpha < .1] = -1
        depth = t_values[torch.arange(t_values.shape[0]), pick]
    return color, alpha, depth


def ground_truth(colors: torch.Tensor, alphas: Optional[torch.Tensor],
                 rays: torch.Tensor, dilate_mode: bool = False):
    """ImageDataset.render.  Restates image_dataset.py:244-262."""
    c = colors[rays]
    if alphas is None or dilate_mode:
        return c, None
    a = alphas[rays]
    c = torch.where(a.unsqueeze(1) > 0, c, torch.zeros_like(c))
    return c, a


def mse_loss(color: torch.Tensor, alpha: torch.Tensor, gt_color: torch.Tensor,
             gt_alpha: Optional[torch.Tensor], alpha_weight: float = 0.1) -> torch.Tensor:
    """ImageDataset.loss.  Restates image_dataset.py:224-242."""
    loss = (gt_color - color).square().mean()
    if alpha_weight > 0 and gt_alpha is not None:
        loss = loss + alpha_weight * (gt_alpha - alpha).square().mean()
    return loss


# --------------------------------------------------------------------------- #
#  a14: optimiser step pieces
# --------------------------------------------------------------------------- #

def lr_decay(initial_lr: float, step: int, decay_rate: float, decay_steps: float) -> float:
    """Restates utils.py:422-445 (exponential_lr_decay), in Python floats."""
    return initial_lr * decay_rate ** (step / decay_steps)


def clip_gradients(grads: List[torch.Tensor], clip_value: float = 0.1,
                   max_norm: float = 0.1) -> float:
    """clip_grad_value_ then clip_grad_norm_ in place; returns the pre-scale norm.

    Restates the calls at ray_caster.py:327-328 with torch's documented
    arithmetic: clamp to [-v, v]; total = ||all grads||_2;
    coef = min(1, max_norm / (total + 1e-6)); grads *= coef.
    """
    for g in grads:
        g.clamp_(-clip_value, clip_value)
    total = torch.linalg.vector_norm(
        torch.stack([torch.linalg.vector_norm(g, 2) for g in grads]), 2)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return float(total)


def adam_update(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor,
                exp_avg_sq: torch.Tensor, step: int, lr: float,
                weight_decay: float = 0.0, beta1: float = 0.9, beta2: float = 0.999,
                eps: float = 1e-8) -> None:
    """One torch.optim.Adam (L2 weight decay, not AdamW) update in place.

    ``step`` is the 1-based count of updates including this one.  Restates the
    optimiser used at ray_caster.py:288,329 (third-party: torch.optim.Adam,
    single-tensor path): g += wd*p; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
    p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps).
    """
    if weight_decay != 0:
        grad = grad + weight_decay * param
    exp_avg.lerp_(grad, 1 - beta1)
    exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (exp_avg_sq.sqrt() / math.sqrt(bc2)).add_(eps)
    param.addcdiv_(exp_avg, denom, value=-(lr / bc1))


# --------------------------------------------------------------------------- #
#  a6 / a15: image assembly and index modes
# --------------------------------------------------------------------------- #

def to_image(valid_local: np.ndarray, colors: np.ndarray, width: int,
             height: int) -> np.ndarray:
    """Scatter ray colours into an (H,W,3) uint8 image.

    Restates ray_sampler.py:191-196: zeros, scatter, ``(x*255).astype(uint8)``
    (truncation, no clipping, no rounding).
    """
    px = np.zeros((height * width, 3), np.float32)
    px[valid_local] = colors
    return (px.reshape(height, width, 3) * 255).astype(np.uint8)


def rgb_to_ycrcb_u8(rgb: np.ndarray) -> np.ndarray:
    """8-bit RGB -> YCrCb the way ``cv2.cvtColor(color, cv2.COLOR_RGB2YCrCb)`` computes it
    (call site image_dataset.py:114-115).

    Third-party algorithm: OpenCV (``opencv-python`` is unpinned in requirements.txt:3 and
    absent from this image), restated from its published 8-bit fixed-point path
    (imgproc colour conversions, ``yuv_shift = 14``):
        Y  = descale(R*4899 + G*9617 + B*1868)          # 0.299, 0.587, 0.114
        Cr = descale((R - Y)*11682 + 128*2^14)          # 0.713, delta = 128
        Cb = descale((B - Y)*9241  + 128*2^14)          # 0.564
    with ``descale(x) = (x + 2^13) >> 14`` and a saturating cast to uint8.
    **Parity unpinned**: the reference's tests hold no vector for it and cv2 cannot be
    imported here; pinned only against the known answers OpenCV's documentation implies
    (tests/test_oracle_golden.py::test_ycrcb_known_answers).  Written with plain Python
    integers per pixel channel on purpose: an independent statement from the product's
    vectorised host code.
    """
    flat = np.asarray(rgb, np.uint8).reshape(-1, 3)
    out = np.empty_like(flat)
    for i, (r, g, b) in enumerate(flat.tolist()):
        y = (r * 4899 + g * 9617 + b * 1868 + 8192) >> 14
        cr = ((r - y) * 11682 + 128 * 16384 + 8192) >> 14
        cb = ((b - y) * 9241 + 128 * 16384 + 8192) >> 14
        out[i] = [min(max(v, 0), 255) for v in (y, cr, cb)]
    return out.reshape(np.asarray(rgb).shape)


def ycrcb_to_rgb_u8(ycrcb: np.ndarray) -> np.ndarray:
    """8-bit YCrCb -> RGB like ``cv2.cvtColor(pixels, cv2.COLOR_YCrCB2RGB)`` (call sites
    ray_sampler.py:197-198, ray_dataset.py:180-181).  OpenCV's published fixed-point inverse:
        R = Y + descale((Cr-128)*22987)                          # 1.403
        G = Y + descale((Cb-128)*(-5636) + (Cr-128)*(-11698))    # -0.344, -0.714
        B = Y + descale((Cb-128)*29049)                          # 1.773
    saturated to uint8.  Parity unpinned (see ``rgb_to_ycrcb_u8``)."""
    flat = np.asarray(ycrcb, np.uint8).reshape(-1, 3)
    out = np.empty_like(flat)
    for i, (y, cr, cb) in enumerate(flat.tolist()):
        cr -= 128
        cb -= 128
        r = y + ((cr * 22987 + 8192) >> 14)
        g = y + ((cb * -5636 + cr * -11698 + 8192) >> 14)
        b = y + ((cb * 29049 + 8192) >> 14)
        out[i] = [min(max(v, 0), 255) for v in (r, g, b)]
    return out.reshape(np.asarray(ycrcb).shape)


def crop_points(width: int, height: int) -> np.ndarray:
    """Pixel ids of the central half-size crop.  Restates image_dataset.py:77-90."""
    res = np.array([width, height], np.float32)
    start = res // 4
    end = res - start
    pts = pixel_grid(width, height)
    inside = ((pts >= start) & (pts < end)).all(-1)
    return np.nonzero(inside)[0]


def sparse_points(width: int, height: int, sparse_size: int) -> np.ndarray:
    """Pixel ids of the sparse grid.  Restates image_dataset.py:473-482."""
    nx = sparse_size * width // height
    ny = sparse_size
    xs = (np.linspace(0, width - 1, nx) + 0.5).astype(np.int32)
    ys = (np.linspace(0, height - 1, ny) + 0.5).astype(np.int32)
    xs, ys = np.meshgrid(xs, ys)
    return ys.reshape(-1) * width + xs.reshape(-1)


# --------------------------------------------------------------------------- #
#  whole training step (used by tests and by bench.py's cpu_baseline leg)
# --------------------------------------------------------------------------- #

class OracleFourierMLP:
    """Parameter holder + autograd forward for the FourierFeatureMLP family."""

    def __init__(self, a_values, b_values, weights, biases):
        self.a_values = a_values
        self.b_values = b_values
        self.weights = [w.clone().requires_grad_(True) for w in weights]
        self.biases = [b.clone().requires_grad_(True) for b in biases]
        self.use_view = False

    def parameters(self) -> List[torch.Tensor]:
        out = []
        for w, b in zip(self.weights, self.biases):
            out += [w, b]
        return out

    def __call__(self, positions, views=None):
        return fourier_mlp_forward(positions, self.a_values, self.b_values,
                                   self.weights, self.biases)


class OracleNeRF:
    """Parameter holder + autograd forward for the full NeRF topology."""

    def __init__(self, params: Dict[str, torch.Tensor], skips, include_inputs):
        self.p = {}
        for key, value in params.items():
            trainable = not key.endswith("encoding")
            self.p[key] = value.clone().requires_grad_(trainable)
        self.skips = list(skips)
        self.include_inputs = include_inputs
        self.use_view = True

    def parameters(self) -> List[torch.Tensor]:
        return [v for k, v in self.p.items() if v.requires_grad]

    def __call__(self, positions, views):
        return nerf_forward(positions, views, self.p, self.skips, self.include_inputs)


class OracleTrainer:
    """zero_grad -> render -> loss -> backward -> clip -> Adam, as ray_caster.py:319-329."""

    def __init__(self, model, lr: float, weight_decay: float = 0.0):
        self.model = model
        self.lr0 = lr
        self.weight_decay = weight_decay
        self.count = 0
        self.m = [torch.zeros_like(p) for p in model.parameters()]
        self.v = [torch.zeros_like(p) for p in model.parameters()]

    def loss(self, positions, views, t_values, gt_color, gt_alpha, alpha_weight=0.1):
        rays, count = t_values.shape
        flat = positions.reshape(-1, 3)
        if self.model.use_view:
            logits = self.model(flat, views.reshape(-1, 3))
        else:
            logits = self.model(flat)
        color, alpha, _ = render(logits.reshape(rays, count, 4), t_values, True)
        return mse_loss(color, alpha, gt_color, gt_alpha, alpha_weight)

    def step(self, positions, views, t_values, gt_color, gt_alpha, lr: float,
             alpha_weight=0.1) -> float:
        params = self.model.parameters()
        for p in params:
            p.grad = None
        loss = self.loss(positions, views, t_values, gt_color, gt_alpha, alpha_weight)
        loss.backward()
        grads = [p.grad for p in params]
        clip_gradients(grads)
        self.count += 1
        with torch.no_grad():
            for p, g, m, v in zip(params, grads, self.m, self.v):
                adam_update(p, g, m, v, self.count, lr, self.weight_decay)
        return float(loss.detach())
