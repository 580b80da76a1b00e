"""Differentiable volumetric raycaster + trainer behind the reference's ``Raycaster`` surface
(ray_caster.py:36-377).

``render`` is an autograd-visible composition of the fused MLP and the composite kernel.
``fit`` keeps the reference's loop structure (LR schedule, epoch shuffles, crop curriculum,
validation cadence, log format) but runs each optimisation step through ``TrainEngine``:
sampling -> fused forward -> composite -> loss -> composite backward -> dgrad/wgrad ->
(RCCL all-reduce) -> fused clip+Adam, all on one HIP stream over flat fp32 buffers, without
Python-side autograd and without host synchronisation.
"""

import copy
import time
from typing import List, NamedTuple, Optional, OrderedDict

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .dataset import RayDataset
from .occupancy import OccupancyGrid
from .sampler import RaySampler, RaySamples
from .utils import RenderResult, check_color_space, learning_rate_at

LogEntry = NamedTuple("LogEntry", [("step", int), ("timestamp", float),
                                   ("state", OrderedDict[str, torch.Tensor]),
                                   ("train_psnr", float), ("val_psnr", float)])


class _Composite(torch.autograd.Function):
    """sigmoid / softplus + front-to-back compositing (kernels K5 / K5b)."""

    @staticmethod
    def forward(ctx, logits, t_values, include_depth, nan_flag):
        logits = logits.contiguous()
        t_values = t_values.contiguous()
        color, alpha, depth = ops.composite_fwd(logits, t_values, include_depth, nan_flag)
        ctx.save_for_backward(logits, t_values)
        if depth is None:
            depth = torch.empty((0,), device=logits.device)
        ctx.mark_non_differentiable(depth)
        return color, alpha, depth

    @staticmethod
    def backward(ctx, d_color, d_alpha, _d_depth):
        logits, t_values = ctx.saved_tensors
        d_logits = ops.composite_bwd(logits, t_values, d_color.contiguous(), d_alpha.contiguous())
        return d_logits, None, None, None


class TrainEngine:
    """One optimisation step as a straight line of kernel launches over flat buffers."""

    def __init__(self, model: nn.Module, weight_decay: float = 0.0, process_group=None,
                 max_samples_per_launch: int = 1 << 22):
        """``max_samples_per_launch`` bounds the activation slabs kept for backward (saved
        activations + dZ: 8.3 KB per sample for the tiny NeRF, 21 KB for the full one, i.e.
        34 / 88 GB at the default of 2^22 samples -- whatever the batch size): larger batches
        run as several forward/backward launches whose gradients are summed before the (single)
        all-reduce and optimiser step -- numerically the same step.  2^22 samples are 128 blocks
        of 32 per wavefront of the persistent kernels: launch and tail effects stay under 1 %
        (the headline batch, 65 536 rays x 64 samples, is exactly one launch)."""
        self.model = model
        self.max_samples = int(max_samples_per_launch)
        params = model._dense_params()
        device = params[0].device
        if device.type != "cuda":
            raise RuntimeError("training runs on the HIP kernels only; move the model to a GPU")
        total = sum(p.numel() for p in params)
        flat = torch.empty((total,), dtype=torch.float32, device=device)
        offset = 0
        for p in params:                      # nn.Parameters become views of one buffer
            n = p.numel()
            flat[offset:offset + n].copy_(p.data.reshape(-1))
            p.data = flat[offset:offset + n].view(p.shape)
            offset += n
        self.flat = flat
        # gradients + the two loss sums travel as ONE buffer, so that data parallel costs one
        # collective per step (the all-reduce of ~1-2.4 MB is latency-bound)
        self.reduce_buf = torch.zeros((total + 2,), dtype=torch.float32, device=device)
        self.grads = self.reduce_buf[:total]
        self.loss_sums = self.reduce_buf[total:]
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.scratch = torch.empty(((total + 1023) // 1024,), dtype=torch.float32, device=device)
        self.grad_norm = torch.zeros((1,), dtype=torch.float32, device=device)
        self.nan_flag = torch.zeros((1,), dtype=torch.int32, device=device)
        self.weight_decay = weight_decay
        self.count = 0
        self.device = device
        self.group = process_group
        self._host_staged = False
        if process_group is not None:
            import torch.distributed as dist
            # a gloo group (CPU tests, several ranks sharing one GPU) reduces through the host
            self._host_staged = dist.get_backend(process_group) == "gloo"
        self.collective_events = None     # set to [] to record (issue, sampling enqueued, behind the wait) events per all-reduce
        # OPT-IN empty-space skipping during training (BASELINE config 5; no counterpart in the
        # reference): with an OccupancyGrid here, the MLP forward / backward run only on the
        # samples in occupied cells; the others are treated as sigma = 0 constants (no colour, no
        # gradient).  New semantics -- PSNR-level parity with the full step, never the default.
        self.occupancy = None
        self.last_evaluated_fraction = None   # device scalar-free diagnostic: M / N of the last launch
        self._saved = {}
        self.loss_history = None          # set to [] to record every step's loss (device scalars)
        model.invalidate_packed()
        assert model.program().num_grad_floats == total

    # ------------------------------------------------------------------ pieces
    def _saved_buffer(self, prog, n):
        need = prog.saved_floats(n)
        buf = self._saved.get("buf")
        if buf is None or buf.numel() < need:
            # grown with headroom (the batch's sample count wobbles from step to step when rays
            # are filtered or empty space is skipped, and every multi-GB allocation costs ~100 ms
            # of page-table work); the old buffer is released first
            self._saved["buf"] = buf = None
            blocks = prog.plan_blocks(n)
            buf = torch.empty((prog.saved_floats(32 * (blocks + blocks // 16 + 1)),),
                              dtype=torch.float32, device=self.device)
            self._saved["buf"] = buf
        return buf

    def _samples(self, sampler: RaySampler, rays: torch.Tensor, step: Optional[int]):
        t, pos, views = sampler.sample_points(rays, step, want_views=self.model.use_view)
        return t, pos.view(-1, 3), None if views is None else views.view(-1, 3)

    def shard(self, rays: torch.Tensor) -> torch.Tensor:
        """This rank's contiguous slice of the (already valid-filtered) global batch."""
        if self.group is None:
            return rays
        import torch.distributed as dist
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        per = -(-rays.numel() // world)
        return rays[rank * per:(rank + 1) * per].contiguous()

    # ------------------------------------------------------------------ train / eval
    def _can_prefetch(self, sampler) -> bool:
        """Samples of the NEXT step may be drawn before this step's optimiser update only when
        they do not depend on the weights (a live coarse model may be the model in training) and
        nothing else reshapes the batch (occupancy compaction keeps its own launch order)."""
        return self.occupancy is None and (not sampler.focus_sampling or sampler.cdfs is not None)

    def _prefetch(self, lookahead):
        """``lookahead`` = (dataset, filtered global ray ids, step) of the NEXT ``train_step``:
        its t-values / positions / view directions (this rank's shard, first launch) are computed
        now -- they depend on the ray state and the noise generator only -- so that in data
        parallel the kernels run UNDER the gradient all-reduce instead of behind it."""
        dataset, next_rays, next_step = lookahead
        sampler = dataset.sampler
        if next_rays is None or next_rays.numel() == 0 or not self._can_prefetch(sampler):
            return
        mine = self.shard(next_rays)
        per_launch = max(1, self.max_samples // sampler.num_samples)
        chunk = mine[:per_launch]
        if mine.numel() == 0:      # (a rank whose shard of the next batch is empty has nothing to sample)
            return
        # the announced tensor itself is kept: `train_step` accepts the look-ahead only for the very
        # same object (an address + count identity would also match a NEW batch allocated where a
        # freed one lay)
        self._prefetched = {"rays": next_rays, "step": next_step, "sampler": sampler,
                            "shard": mine, "samples": self._samples(sampler, chunk, next_step)}

    def train_step(self, dataset, batch, step: int, lr: float,
                   rays: Optional[torch.Tensor] = None, lookahead=None) -> torch.Tensor:
        """zero_grad -> loss -> backward -> clip value -> clip norm -> Adam, as
        ray_caster.py:319-329.  Returns the (global) batch loss as a device scalar.
        ``rays`` = the already filtered global ray ids of ``batch`` (see
        ``RayDataset.epoch_ray_ids``), which saves the per-step device-to-host sync.
        ``lookahead`` = (dataset, rays, step) of the next call: see ``_prefetch``."""
        sampler = dataset.sampler
        all_rays = dataset.ray_ids(batch) if rays is None else rays
        global_count = int(all_rays.numel())
        ahead, self._prefetched = getattr(self, "_prefetched", None), None
        if ahead is not None and not (ahead["rays"] is all_rays and ahead["step"] == step and
                                      ahead["sampler"] is sampler):
            ahead = None          # (not the batch that was announced: sample afresh)
        rays = self.shard(all_rays) if ahead is None else ahead["shard"]
        count = int(rays.numel())
        alphas = dataset._gt_alphas()
        aw = float(dataset.alpha_weight) if alphas is not None else 0.0
        prog = self.model.program()
        sums = self.loss_sums
        per_launch = max(1, self.max_samples // sampler.num_samples)
        if count == 0:
            self.reduce_buf.zero_()
        precision = getattr(self.model, "train_precision", "f32")
        if hasattr(self.model, "effective_precision"):
            precision = self.model.effective_precision(precision)
        for lo in range(0, count, per_launch):
            chunk = rays[lo:lo + per_launch]
            first = lo == 0
            t, pos, views = ahead["samples"] if (first and ahead is not None) else self._samples(sampler, chunk, step)
            index = None
            if self.occupancy is not None:
                # compaction (one D2H sync for the count), MLP on the occupied samples only
                total = pos.shape[0]
                pos, views, index = self.occupancy.compact(pos, views)
                self.last_evaluated_fraction = pos.shape[0] / max(total, 1)
                saved = self._saved_buffer(prog, max(pos.shape[0], 1))
                packed = prog.forward(pos, views, saved, precision=precision)
                logits = ops.scatter_logits(packed, index, total)
            else:
                saved = self._saved_buffer(prog, pos.shape[0])
                logits = prog.forward(pos, views, saved, precision=precision)
            # composite forward, ground-truth gather + loss sums, composite backward: one launch
            d_logits, partials = ops.composite_train(logits, t, dataset.colors, alphas, chunk,
                                                     1.0 / (3 * global_count), aw / global_count,
                                                     self.nan_flag)
            d_logits = d_logits.view(-1, 4)
            # one launch on one device: the loss comes straight from the partial sums at the end
            solo = self.group is None and count <= per_launch
            part = None
            if not solo:
                part = sums if first else torch.empty_like(sums)
                ops.loss_from_partials(partials, global_count, aw, sums_out=part, want_loss=False)
            if index is not None:
                d_logits = ops.gather_logits(d_logits, index)
            if first:
                prog.backward(d_logits, pos, views, saved, self.grads, precision=precision)
            else:
                if getattr(self, "_grads_part", None) is None:
                    self._grads_part = torch.empty_like(self.grads)
                prog.backward(d_logits, pos, views, saved, self._grads_part, precision=precision)
                self.grads.add_(self._grads_part)
                sums.add_(part)
        if self.group is not None:
            self._all_reduce(lookahead)
        elif lookahead is not None:
            self._prefetch(lookahead)
        if global_count == 0:
            # no ray of the batch hits the volume (every rank sees the same global count).  The
            # reference takes the mean of empty tensors here (ray_caster.py:321-326): a NaN loss
            # whose backward poisons every weight.  The loss is reported the same way (0 / 0), but
            # the optimiser step is skipped: no moment decay, no weight decay, weights intact.
            loss = ops.loss_value(sums, 0, aw)
            if self.loss_history is not None:       # (one entry per call, skipped step or not)
                self.loss_history.append(loss)
            return loss
        self.count += 1
        ops.clip_adam(self.flat, self.grads, self.exp_avg, self.exp_avg_sq, self.count, lr,
                      weight_decay=self.weight_decay, scratch=self.scratch,
                      norm_out=self.grad_norm)
        self.model.invalidate_packed()
        # (a fresh tensor: the reduce buffer is overwritten by the next step)
        if self.group is None and count <= per_launch:
            loss = ops.loss_from_partials(partials, global_count, aw)
        else:
            loss = ops.loss_value(sums, global_count, aw)
        if self.loss_history is not None:
            self.loss_history.append(loss)
        return loss

    def _all_reduce(self, lookahead=None):
        """Sum of [flat gradients | 2 loss sums] over the ranks: one RCCL all-reduce over xGMI
        (or, for a gloo group, one staged through the host).  The RCCL collective is issued
        asynchronously -- it runs on the communicator's own stream, behind everything enqueued so
        far -- and the launch stream waits for it only in front of the optimiser kernel: the
        sampling kernels of the NEXT step (``lookahead``), which need neither gradients nor
        weights, run under it.  At the reference's default batch (a 1.2 ms step) that hides
        ~15 us of sampling behind the 30-50 us latency-bound collective."""
        import torch.distributed as dist
        events = self.collective_events
        if events is not None:
            # three stamps on the LAUNCH stream: issue | the look-ahead sampling kernels enqueued |
            # behind the wait.  (e0, e1) spans max(collective, sampling), NOT the collective alone;
            # (e0, em) is the sampling that ran under it, and what the step really pays for the
            # collective is the rest, (em, e1): the time the launch stream stood waiting.
            e0, em, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
        if self._host_staged:
            host = self.reduce_buf.cpu()
            dist.all_reduce(host, group=self.group)
            self.reduce_buf.copy_(host)
            if events is not None:
                em.record()          # (host-staged: nothing overlaps, the sampling comes behind)
            if lookahead is not None:
                self._prefetch(lookahead)
        else:
            work = dist.all_reduce(self.reduce_buf, group=self.group, async_op=True)
            if lookahead is not None:
                self._prefetch(lookahead)
            if events is not None:
                em.record()
            work.wait()              # (the launch stream waits; the host does not)
        if events is not None:
            e1.record()
            events.append((e0, em, e1))

    def reduce_small(self, values: torch.Tensor) -> torch.Tensor:
        """Sum of a few device floats over the ranks (sharded validation): in place."""
        if self.group is None:
            return values
        import torch.distributed as dist
        if self._host_staged:
            host = values.cpu()
            dist.all_reduce(host, group=self.group)
            values.copy_(host)
        else:
            dist.all_reduce(values, group=self.group)
        return values

    def shared_seed(self, draw) -> int:
        """``draw()`` evaluated on rank 0 (a non-negative int < 2^62) and made known to every
        rank with one 8-byte broadcast -- what data parallel needs to walk ONE permutation
        without shipping it ("the shuffle permutation must be generated from a
        shared seed")."""
        if self.group is None:
            return int(draw())
        import torch.distributed as dist
        root = dist.get_global_rank(self.group, 0)
        value = int(draw()) if dist.get_rank(self.group) == 0 else 0
        seed = torch.tensor([value], dtype=torch.int64,
                            device="cpu" if self._host_staged else self.device)
        dist.broadcast(seed, root, group=self.group)
        return int(seed.item())

    def eval_loss(self, dataset, batch, step: Optional[int]) -> torch.Tensor:
        """Forward-only loss of a batch (the body of ``_validate``)."""
        sampler = dataset.sampler
        rays = dataset.ray_ids(batch)
        count = int(rays.numel())
        if count == 0:      # mean over an empty batch: NaN like the reference, not an error
            return torch.full((), float("nan"), dtype=torch.float32, device=self.device)
        alphas = dataset._gt_alphas()
        aw = float(dataset.alpha_weight) if alphas is not None else 0.0
        t, pos, views = self._samples(sampler, rays, step)
        # (a forward-only pass: the model's INFERENCE arithmetic, like a no-grad model call)
        mode = getattr(self.model, "precision", "f32")
        if hasattr(self.model, "effective_precision"):
            mode = self.model.effective_precision(mode)
        prog = self.model.program()
        logits = prog.forward16(pos, views) if mode == "bf16x3" else prog.forward(pos, views, None, precision=mode)
        color, alpha, _ = ops.composite_fwd(logits, t, False, self.nan_flag)
        sums, _, _ = ops.mse_loss(color, alpha, dataset.colors, alphas, rays, 1.0, 1.0,
                                  want_grad=False)
        return ops.loss_value(sums, count, aw)

    def check_finite(self):
        """Raises like the asserts at ray_caster.py:73-74 (checked lazily: one sync)."""
        assert int(self.nan_flag.item()) == 0, "NaN in sigmoid(rgb) / softplus(sigma)"


class Raycaster(nn.Module):
    """Volumetric raycaster around a radiance-field model."""

    def __init__(self, model: nn.Module):
        nn.Module.__init__(self)
        self.model = model
        self._nan_flag = None
        # epoch permutations: "numpy" = np.random.shuffle on the host like the reference
        # (ray_caster.py:312-313); "device" = torch.randperm on the GPU from torch's generator;
        # "seeded" = one seed drawn from np.random per epoch, permutation by a device randperm
        # from that seed.  Data parallel always runs "seeded" with rank 0's seed (8 bytes
        # broadcast per epoch instead of the 128-512 MB permutation itself).
        self.shuffle_source = "numpy"
        self.process_group = None         # set to a torch.distributed group for data parallel
        self.occupancy = None             # an OccupancyGrid switches on empty-space skipping (no_grad renders)
        self.fused_render = True          # render_image through the one-launch kernel (False: never; "always": see _can_fuse)
        # OPT-IN empty-space skipping DURING `fit` (new semantics, DESIGN K9; BASELINE config 5):
        # (warm-up steps, refresh interval) -- after the warm-up of exact steps the training
        # engine gets an occupancy grid derived from the model itself, rebuilt every interval
        self.train_occupancy_schedule = None
        self.train_occupancy_resolution = 128
        self.train_occupancy_threshold = 0.01

    # ------------------------------------------------------------------ rendering
    def _flag(self, device):
        if self._nan_flag is None or self._nan_flag.device != device:
            self._nan_flag = torch.zeros((1,), dtype=torch.int32, device=device)
        return self._nan_flag

    def check_finite(self):
        if self._nan_flag is not None:
            assert int(self._nan_flag.item()) == 0, "NaN in sigmoid(rgb) / softplus(sigma)"

    def render(self, ray_samples: RaySamples, include_depth=False) -> RenderResult:
        """Per-ray colour / alpha / depth of the samples; differentiable w.r.t. the model
        (ray_caster.py:48-93)."""
        num_rays, num_samples = ray_samples.positions.shape[:2]
        positions = ray_samples.positions.reshape(-1, 3)
        views = ray_samples.view_directions.reshape(-1, 3) if self.model.use_view else None
        if self.occupancy is not None and not torch.is_grad_enabled() and positions.shape[0] > 0:
            # opt-in empty-space skipping (inference only): the MLP sees the occupied samples
            pos_c, view_c, index = self.occupancy.compact(positions.contiguous(),
                                                          None if views is None else views.contiguous())
            if pos_c.shape[0] > 0:
                packed = self.model(pos_c, view_c) if views is not None else self.model(pos_c)
            else:
                packed = torch.empty((0, 4), dtype=torch.float32, device=positions.device)
            logits = ops.scatter_logits(packed.contiguous(), index, positions.shape[0])
        elif views is not None:
            logits = self.model(positions, views)
        else:
            logits = self.model(positions)
        logits = logits.reshape(num_rays, num_samples, 4)
        color, alpha, depth = _Composite.apply(logits, ray_samples.t_values, include_depth,
                                               self._flag(logits.device))
        return RenderResult(color, alpha, depth if include_depth else None)

    def _loss(self, step: int, dataset: RayDataset, batch) -> torch.Tensor:
        device = next(self.model.parameters()).device
        rays = dataset.get_rays(batch, step).to(device)
        return dataset.loss(step, rays, self.render(rays, True))

    def batched_render(self, samples: RaySamples, batch_size: int,
                       include_depth: bool) -> RenderResult:
        """Renders in batches without gradients; returns numpy arrays
        (ray_caster.py:103-138)."""
        self.model.eval()
        colors, alphas, depths = [], [], []
        with torch.no_grad():
            device = next(self.model.parameters()).device
            total = len(samples.positions)
            for start in range(0, total, batch_size):
                end = min(start + batch_size, total)
                pred = self.render(samples.subset(slice(start, end)).to(device), include_depth)
                colors.append(pred.color)
                alphas.append(pred.alpha)
                if include_depth:
                    depths.append(pred.depth)
        self.model.train()
        out = RenderResult(torch.cat(colors), torch.cat(alphas),
                           torch.cat(depths) if include_depth else None).numpy()
        self.check_finite()
        return out

    @staticmethod
    def _inference_mode(model) -> str:
        mode = getattr(model, "precision", "f32")
        return model.effective_precision(mode) if hasattr(model, "effective_precision") else mode

    def _can_fuse(self, sampler: RaySampler) -> bool:
        model = self.model
        if not (self.fused_render and hasattr(model, "program") and sampler.num_samples <= 256
                and self._inference_mode(model) == "f32"):
            return False
        # 512-wide chains: the pair-of-waves variant equals the three-pass rate without a grid
        # (1.33 vs 1.35 frames/s at 800x800x128) but loses to the globally compacted K9 path
        # with one (14.3 vs 18.6: per-ray blocks of 32 and two pairs in step), so that case
        # keeps the three passes
        # (``fused_render = "always"`` overrides, for measurements)
        if model.program().big:         # layers wider than 512 channels: no fused render kernel
            return False
        return self.fused_render == "always" or not (model.program().wide and self.occupancy is not None)

    def render_rays(self, sampler: RaySampler, rays, include_depth=False,
                    image: Optional[torch.Tensor] = None, pixel_offset: int = 0,
                    want_color=True) -> RenderResult:
        """Inference render of the sampler's rays ``rays`` (device int64 ids, or a
        ``(first id, count)`` range that the kernel filters by the validity mask) in one launch
        of the fused kernel: sampling, encoding, MLP and compositing without materialising
        samples or logits (what ``sampler.sample`` + ``render`` do in four passes over HBM,
        ray_caster.py:103-138).  Stratified / opacity-guided samplers contribute their
        t-values; a plain uniform sampler needs none."""
        needs_t = sampler.stratified or sampler.focus_sampling
        if isinstance(rays, tuple):
            spec = (int(rays[0]), int(rays[1]), sampler.valid)
            index = torch.arange(spec[0], spec[0] + spec[1], dtype=torch.int64,
                                 device=sampler.device) if needs_t else None
        else:
            spec = index = rays.contiguous()
        t_values = sampler.sample_t(index, None) if needs_t else None
        color, alpha, depth = self.model.program().render(
            sampler.starts, sampler.directions, sampler.near_far, spec,
            sampler.num_samples, sampler._unit(sampler.num_samples), t_values,
            occupancy=self.occupancy, want_color=want_color, want_depth=include_depth,
            nan_flag=self._flag(sampler.device), image=image, pixel_offset=pixel_offset)
        return RenderResult(color, alpha, depth)

    def render_image_device(self, sampler: RaySampler, index: int, batch_size: int,
                            color_space="RGB") -> torch.Tensor:
        """(H,W,3) uint8 RGB frame of camera ``index % num_cameras`` as a DEVICE tensor, enqueued
        without any host synchronisation (callers overlap the copy-out / encoding, see
        ``frames.FrameSink``).  ``color_space`` names the space the MODEL predicts in ("YCrCb":
        the u8 frame goes through kernel K8b like ray_sampler.py:197-198)."""
        check_color_space(color_space)
        camera = index % sampler.num_cameras
        self.model.eval()
        with torch.no_grad():
            if self._can_fuse(sampler):
                image = torch.zeros((sampler.image_height, sampler.image_width, 3),
                                    dtype=torch.uint8, device=sampler.device)
                first = camera * sampler.rays_per_camera
                self.render_rays(sampler, (first, sampler.rays_per_camera), False, image=image,
                                 pixel_offset=first, want_color=False)
            else:
                rays = sampler._valid_for_camera(camera)
                colors = torch.empty((rays.numel(), 3), dtype=torch.float32, device=sampler.device)
                for start in range(0, rays.numel(), batch_size):
                    chunk = sampler.sample(rays[start:start + batch_size].contiguous(), None)
                    colors[start:start + batch_size] = self.render(chunk, False).color
                image = ops.to_image(colors, (rays - camera * sampler.rays_per_camera).contiguous(),
                                     sampler.image_width, sampler.image_height)
        if color_space == "YCrCb":
            ops.ycrcb_to_rgb_u8(image)
        self.model.train()
        return image

    def render_image(self, sampler: RaySampler, index: int, batch_size: int,
                     color_space="RGB") -> np.ndarray:
        """(H,W,3) uint8 frame of camera ``index % num_cameras`` (ray_caster.py:140-159)."""
        image = self.render_image_device(sampler, index, batch_size, color_space)
        self.check_finite()
        return image.cpu().numpy()

    def render_activations(self, *args, **kwargs):
        raise NotImplementedError("render_activations is a lecture visualisation outside the "
                                  "HIP hot path")

    def to_scenepic(self, *args, **kwargs):
        raise NotImplementedError("scenepic export is outside the HIP hot path")

    # ------------------------------------------------------------------ training
    def _validate(self, engine: TrainEngine, dataset: RayDataset, batch_size: int,
                  step: int) -> float:
        """PSNR = -10 log10(mean batch loss) over <= 102400 evenly spaced rays, full batches
        only (ray_caster.py:220-246)."""
        num_rays = len(dataset)
        num_validate = min(num_rays, 1024 * 100)
        if num_validate < num_rays:
            index = np.linspace(0, num_rays, num_validate, endpoint=False).astype(np.int32)
            index = np.asarray(dataset.to_valid(index.tolist()), np.int64)
        else:
            index = np.arange(num_rays)
        index = torch.from_numpy(np.asarray(index, np.int64)).to(engine.device)
        starts = [s for s in range(0, num_validate, batch_size) if s + batch_size <= len(index)]
        self.model.eval()
        if not starts:                       # tiny dataset: one short batch instead of a crash
            mean = float(engine.eval_loss(dataset, index, step).item())
        else:
            # data parallel: the batches are dealt round-robin to the ranks and the partial sums
            # meet in one tiny all-reduce (every rank reports the same PSNR)
            mine = starts
            if engine.group is not None:
                import torch.distributed as dist
                mine = starts[dist.get_rank(engine.group)::dist.get_world_size(engine.group)]
            losses = [engine.eval_loss(dataset, index[s:s + batch_size], step) for s in mine]
            total = torch.stack(losses).sum().reshape(1) if losses else \
                torch.zeros((1,), dtype=torch.float32, device=engine.device)
            mean = float(engine.reduce_small(total).item()) / len(starts)
        self.model.train()
        return float(-10. * np.log10(mean))

    def _epoch_order(self, num_rays: int, engine: TrainEngine) -> torch.Tensor:
        """One epoch's permutation of the dataset-local indices, on the device."""
        source = self.shuffle_source
        if source == "numpy" and self.process_group is not None and \
                torch.distributed.get_world_size(self.process_group) > 1:
            source = "seeded"                # every rank must walk the same permutation
        if source == "numpy":
            order = np.arange(num_rays)
            np.random.shuffle(order)
            return torch.from_numpy(order).to(engine.device)
        if source == "device" and self.process_group is None:
            return torch.randperm(num_rays, device=engine.device)
        if source == "device":
            draw = lambda: int(torch.randint(0, 2 ** 62, (1,)).item())        # noqa: E731
        elif source == "seeded":
            draw = lambda: int(np.random.randint(0, 2 ** 62, dtype=np.int64))  # noqa: E731
        else:
            raise ValueError("shuffle_source is 'numpy', 'device' or 'seeded'")
        generator = torch.Generator(device=engine.device)
        generator.manual_seed(engine.shared_seed(draw))
        return torch.randperm(num_rays, generator=generator, device=engine.device)

    def fit(self, train_dataset: RayDataset, val_dataset: RayDataset, batch_size: int,
            learning_rate: float, num_steps: int, crop_steps: int, report_interval: int,
            decay_rate: float, decay_steps: int, weight_decay: float, visualizers: List,
            disable_aml=False) -> List[LogEntry]:
        """Trains the model; same arguments, schedule and log lines as ray_caster.py:248-377."""
        trainval_dataset = train_dataset.sample_cameras(val_dataset.num_cameras,
                                                        val_dataset.num_samples, False)
        engine = TrainEngine(self.model, weight_decay, self.process_group)
        self.engine = engine
        if self.process_group is not None:       # every rank starts from rank 0's weights
            root = torch.distributed.get_global_rank(self.process_group, 0)
            if engine._host_staged:
                host = engine.flat.cpu()
                torch.distributed.broadcast(host, root, group=self.process_group)
                engine.flat.copy_(host)
            else:
                torch.distributed.broadcast(engine.flat, root, group=self.process_group)
            self.model.invalidate_packed()
        step = 0
        start_time = time.time()
        log = []
        dataset_mode = train_dataset.mode
        if crop_steps:
            for ds in (train_dataset, val_dataset, trainval_dataset):
                ds.mode = RayDataset.Mode.Center
        else:
            val_dataset.mode = dataset_mode
            trainval_dataset.mode = dataset_mode

        def render_image(samples: RaySamples, include_depth: bool):
            return self.batched_render(samples, batch_size, include_depth)

        def render_act(sampler: RaySampler, camera: int):
            return self.render_activations(sampler, camera, batch_size, train_dataset.color_space)

        is_main = self.process_group is None or torch.distributed.get_rank(self.process_group) == 0
        while step <= num_steps:
            num_rays = len(train_dataset)
            order = self._epoch_order(num_rays, engine)
            # valid-ray filter of the whole epoch in one go (one sync per epoch, not per step)
            epoch_rays, bounds = train_dataset.epoch_ray_ids(order, batch_size)
            announced = None
            for bi, start in enumerate(range(0, num_rays, batch_size)):
                if step > num_steps:
                    break
                lr = learning_rate_at(learning_rate, step, decay_rate, decay_steps)
                batch = order[start:min(start + batch_size, num_rays)]
                if self.train_occupancy_schedule is not None:
                    warm, every = self.train_occupancy_schedule
                    if step >= warm and (step - warm) % max(int(every), 1) == 0:
                        # (data parallel: identical weights on every rank give identical grids)
                        engine.occupancy = OccupancyGrid.from_model(
                            self.model, train_dataset.sampler.bounds, self.train_occupancy_resolution,
                            self.train_occupancy_threshold, True)
                # the next batch of the epoch, announced so that its sampling kernels run under
                # this step's gradient all-reduce (data parallel only; never with host-side noise,
                # whose draws a discarded look-ahead -- crop removal, last step -- would shift
                # against the reference's generator)
                # (the engine honours an announcement only for the very tensor it was given)
                mine = announced if announced is not None else epoch_rays[bounds[bi]:bounds[bi + 1]]
                ahead = announced = None
                if (engine.group is not None and bi + 2 < len(bounds) and step < num_steps
                        and train_dataset.sampler.noise_source != "host"):
                    announced = epoch_rays[bounds[bi + 1]:bounds[bi + 2]]
                    ahead = (train_dataset, announced, step + 1)
                engine.train_step(train_dataset, batch, step, lr, rays=mine, lookahead=ahead)

                if step < 10 or step % report_interval == 0:
                    engine.check_finite()
                    train_psnr = self._validate(engine, trainval_dataset, batch_size, step)
                    val_psnr = self._validate(engine, val_dataset, batch_size, step)
                    now = time.time()
                    if step >= report_interval:
                        per_step = (now - start_time) / step
                        eta = time.strftime("%a, %d %b %Y %H:%M:%S +0000",
                                            time.gmtime(now + (num_steps - step) * per_step))
                    else:
                        per_step = 0
                        eta = "N/A"
                    if is_main:
                        print("{:07}".format(step), "{:2f} s/step".format(per_step),
                              "psnr_train: {:2f}".format(train_psnr),
                              "val_psnr: {:2f}".format(val_psnr), "lr: {:.2e}".format(lr),
                              "eta:", eta)
                    if step % report_interval == 0:
                        state = copy.deepcopy(self.model.state_dict())
                        log.append(LogEntry(step, now - start_time, state, train_psnr, val_psnr))
                    if train_dataset.mode == RayDataset.Mode.Center and step >= crop_steps:
                        if is_main:
                            print("Removing center crop...")
                        for ds in (train_dataset, val_dataset, trainval_dataset):
                            ds.mode = dataset_mode
                        step += 1
                        break

                for visualizer in visualizers:
                    visualizer.visualize(step, render_image, render_act)
                step += 1
        return log
