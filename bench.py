"""Benchmark of the NeRF volume-rendering hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" is one complete optimisation step of the reference's training loop
(ray_caster.py:319-329) on one batch of synthetic rays: stratified t-sampling -> fused
Fourier-MLP forward -> alpha compositing -> loss -> backward (composite, dgrad, wgrad) ->
[one RCCL all-reduce] -> clip + Adam.  Workload = BASELINE.json configs[1]: tiny NeRF
(PositionalFourierMLP(3,4,5.5), 256 channels) on a synthetic 100 x 400x400 RGBA dataset,
64 samples/ray, 65536 rays per GPU per step (weak scaling).  Inputs (ray state, ground truth,
weights) are resident in HBM before the timed region.

``--gpus N`` without a launcher spawns its own N ranks (torch.distributed.run, one per GPU).
Rank 0 prints ONE JSON line: the metric, the roofline of the dominant kernel, per-kernel
timings, the all-reduce cost (N > 1), and -- at N = 1 -- the reporting-only legs: frames/sec of
the fused render (with and without frame I/O), the MLP kernels at the north-star launch shape,
a full optimisation step of BASELINE configs[2] (full NeRF, 64 uniform + 64 opacity-guided
samples with a live coarse model) and the CPU baseline.
"""

import argparse
import contextlib
import io
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
TRAFFIC_PROFILE = "profiles/r06_hbm_traffic.json"
SUSTAINED_PROFILE = "profiles/r06_sustained_bf16_matrix_rate.json"

KERNEL_OF = {"ffn_mlp_forward": "mlp_forward_kernel<train>",
             "ffn_mlp_backward_data": "mlp_backward_data_kernel",
             "ffn_mlp_wgrad_units": "wgrad_unit_kernel"}
# (kernel symbols as rocprofv3 prints them; the second template argument of the chain kernels is
# the number of waves per block, the third -- round 5 -- the team-of-four variant of the 1024-wide
# chains: round 4's profiles call the same kernels <1, 1> / <1>, round 3's <1, false> / <false>)
SYMBOL_OF = {"mlp_forward_kernel<train>": ("ffn::mlp_forward_kernel<1, 1, false>", "ffn::mlp_forward_kernel<1, 1>",
                                           "ffn::mlp_forward_kernel<1, false>"),
             "mlp_backward_data_kernel": ("ffn::mlp_backward_data_kernel<1, false>", "ffn::mlp_backward_data_kernel<1>",
                                          "ffn::mlp_backward_data_kernel<false>"),
             "wgrad_unit_kernel": ("ffn::wgrad_unit_kernel",)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rays", type=int, default=65536,
                    help="rays per GPU per step (--scaling weak) / per step of the whole job (strong)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --rays per GPU per step, the global batch grows with N (the driver's "
                         "contract); strong: ONE global batch of --rays rays per step, sharded over "
                         "the N ranks (BASELINE's '1 -> 8 GPU ray throughput' read on a fixed workload)")
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--cameras", type=int, default=100)
    ap.add_argument("--size", type=int, default=400)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-render", action="store_true", help="skip the frames/sec leg")
    ap.add_argument("--no-target-shape", action="store_true",
                    help="skip the MFMA-utilisation leg at the north-star shape (NeRF, 65536 x 128)")
    ap.add_argument("--no-config3", action="store_true",
                    help="skip the full-NeRF + focus-sampling optimisation-step leg")
    ap.add_argument("--no-config5", action="store_true",
                    help="skip the 512-wide Gaussian-feature 800x800 optimisation-step leg")
    ap.add_argument("--no-bf16-leg", action="store_true",
                    help="skip the (separately labelled) split-bf16 inference leg")
    ap.add_argument("--no-skip-leg", action="store_true",
                    help="skip the (separately labelled) empty-space-skipping leg")
    ap.add_argument("--model", default="tiny", choices=["tiny", "nerf", "gaussian512"],
                    help="tiny = BASELINE configs[1] (the metric's config); nerf = configs[2]-shaped "
                         "full NeRF (8x256, skip, view branch), use with --samples 128")
    return ap.parse_args()


def synthetic_rig(num_cameras, size, fov_deg=40.0, distance=4.0, seed=20080524):
    """Cameras on a seeded ring/hemisphere looking at the origin (x right, y down, z fwd)."""
    rng = np.random.RandomState(seed)
    focal = 0.5 * size / np.tan(0.5 * np.deg2rad(fov_deg))
    intr = np.array([[focal, 0, size / 2], [0, focal, size / 2], [0, 0, 1]], np.float32)
    poses = []
    for c in range(num_cameras):
        azi = 2 * np.pi * c / num_cameras
        alt = np.deg2rad(10 + 35 * rng.rand())
        eye = distance * np.array([np.cos(azi) * np.cos(alt), np.sin(alt), np.sin(azi) * np.cos(alt)])
        fwd = -eye / np.linalg.norm(eye)
        right = np.cross(fwd, np.array([0, 1.0, 0]))
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        pose = np.eye(4, dtype=np.float32)
        pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, down, fwd, eye
        poses.append(pose)
    return intr, poses


def analytic_images(sampler, radius=0.6):
    """RGBA uint8 images of a shaded sphere, rendered from the sampler's own ray state."""
    o, d = sampler.starts, sampler.directions
    b = (o * d).sum(-1)
    c = (o * o).sum(-1) - radius * radius
    disc = b * b - c
    hit = disc > 0
    t = -b - torch.sqrt(torch.clamp(disc, min=0))
    normal = torch.nn.functional.normalize(o + t.unsqueeze(-1) * d, dim=-1)
    rgb = (0.5 + 0.5 * normal) * hit.unsqueeze(-1)
    rgba = torch.cat([rgb, hit.unsqueeze(-1).float()], -1)
    img = (rgba * 255).to(torch.uint8).reshape(sampler.num_cameras, sampler.image_height,
                                                sampler.image_width, 4)
    return img.cpu().numpy()


def batch_plan(rays, world, scaling):
    """(rays per step of the whole job, rays per GPU per step).  weak: `rays` per GPU, the global
    batch grows with the ranks (the driver's contract: per-GPU work fixed); strong: `rays` per step
    in all, sharded over the ranks (TrainEngine.shard: contiguous slices of the valid-filtered
    global batch)."""
    if scaling == "weak":
        return rays * world, rays
    if scaling != "strong":
        raise SystemExit("bench.py --scaling is weak or strong")
    if rays % world:
        raise SystemExit("bench.py --scaling strong: --rays %d is not a multiple of %d ranks" % (rays, world))
    return rays, rays // world


def physical_cores():
    """(physical cores, logical CPUs) of this host: distinct (package, core) pairs of
    /proc/cpuinfo, falling back to the logical count."""
    logical = os.cpu_count() or 1
    try:
        pairs, phys, core = set(), None, None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        pairs.add((phys, core))
                    phys = core = None
        if phys is not None and core is not None:
            pairs.add((phys, core))
        if pairs:
            return len(pairs), logical
    except OSError:
        pass
    return logical, logical


def cpu_baseline(args, model_state):
    """The oracle's training step (the reference's ATen op sequence restated) on the host
    cores, on a bounded sample of the same workload: full training step and forward only.
    `cores` = the torch thread count actually used: the FASTEST of a sweep over 8 / 16 / 32 / 64 /
    every physical core (two steps each, shown in `thread_sweep`: torch's CPU kernels stop scaling
    -- and then regress -- far below the core count of a GPU host); a second sample times the
    step at 8 192 rays (closer to the GPU line's 65 536 rays per step) at that thread count."""
    from oracle import ffn_oracle as orc
    phys, logical = physical_cores()
    S = args.samples
    rng = torch.Generator().manual_seed(1)
    ws = [model_state["layers.%d.weight" % i].cpu() for i in range(4)]
    bs = [model_state["layers.%d.bias" % i].cpu() for i in range(4)]
    model = orc.OracleFourierMLP(model_state["a_values"].cpu(), model_state["b_values"].cpu(), ws, bs)
    trainer = orc.OracleTrainer(model, 5e-4)

    def workload(rays):
        near = torch.full((rays,), 3.0)
        far = torch.full((rays,), 5.0)
        starts = torch.randn(rays, 3, generator=rng)
        starts = 4 * starts / starts.norm(dim=-1, keepdim=True)
        dirs = -starts / 4
        gt_c, gt_a = torch.rand(rays, 3, generator=rng), (torch.rand(rays, generator=rng) > 0.4).float()

        def sample():
            noise = torch.rand((rays, S), generator=rng)
            t = orc.uniform_t(near, far, S, noise)
            return starts.unsqueeze(1) + t.unsqueeze(-1) * dirs.unsqueeze(1), t

        def one_step():
            pos, t = sample()
            trainer.step(pos, None, t, gt_c, gt_a, 5e-4)

        def one_forward():
            pos, t = sample()
            trainer.loss(pos, None, t, gt_c, gt_a)

        return one_step, one_forward

    rays = 1024
    one_step, one_forward = workload(rays)
    sweep = {}
    for threads in sorted({t for t in (8, 16, 32, 64, phys) if t <= phys} or {phys}):
        torch.set_num_threads(threads)
        one_step()
        t0 = time.time()
        one_step()
        one_step()
        sweep[threads] = rays * 2 / (time.time() - t0)
    cores = max(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    one_step()
    t0 = time.time()
    done = 0
    while time.time() - t0 < 8.0 and done < 40 or done < 2:
        one_step()
        done += 1
    elapsed = time.time() - t0
    # forward only (sampling + model + compositing, no autograd): the render-side baseline
    t1 = time.time()
    fwd_done = 0
    with torch.no_grad():
        while time.time() - t1 < 4.0 and fwd_done < 40 or fwd_done < 2:
            one_forward()
            fwd_done += 1
    fwd_elapsed = time.time() - t1
    # the same step at 8 192 rays: one warm-up step, then THREE timed ones (one ~2 s step wobbled by
    # 25 % between boxes)
    big_step, _ = workload(8192)
    big_step()
    big_times = []
    for _ in range(3):
        t2 = time.time()
        big_step()
        big_times.append(time.time() - t2)
    big_elapsed = sum(big_times)
    return {"value": rays * done / elapsed, "unit": "rays/s", "cores": cores,
            "physical_cores": phys, "logical_cpus": logical, "kind": "port",
            "thread_sweep_rays_per_s": {str(k): round(v, 1) for k, v in sorted(sweep.items())},
            "cores_chosen_because": "fastest of the sweep (two 1024-ray training steps per thread count)",
            "forward_only_rays_per_s": rays * fwd_done / fwd_elapsed,
            "at_8192_rays_per_step_rays_per_s": round(3 * 8192 / big_elapsed, 1),
            "at_8192_rays_per_step_seconds_per_step": [round(x, 3) for x in big_times],
            "sample": "%d training steps (%.1f s) and %d forward passes (%.1f s) of %d rays x %d "
                      "samples, three more steps of 8192 rays (%.1f s) (oracle: the reference's ATen op "
                      "sequence on the host CPU, torch.set_num_threads(%d))"
                      % (done, elapsed, fwd_done, fwd_elapsed, rays, S, big_elapsed, cores)}


CPU_BASELINE_CACHE = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ffn_bench_cpu_baseline.json")
CPU_BASELINE_MAX_AGE_S = 6 * 3600


def _baseline_key(args):
    import socket
    phys, logical = physical_cores()
    return {"samples": args.samples, "host": socket.gethostname(), "physical_cores": phys,
            "logical_cpus": logical, "commit": git_head(), "uid": os.getuid()}


def cpu_baseline_for(args, model_state, world):
    """The `cpu_baseline` object of a bench line.  N = 1 measures it (and leaves the result in a
    per-box cache file); N > 1 lines carry the same object -- the cached N = 1 measurement of this
    box when the driver ran N = 1 first (same host, core counts, commit and user, at most six hours
    old: anything else is measured again), otherwise a fresh bounded measurement on rank 0 (the other
    ranks wait at the final barrier) -- tagged with where it came from."""
    key = _baseline_key(args)
    if world == 1:
        out = cpu_baseline(args, model_state)
        out["source"] = "measured in this run (N = 1)"
        try:
            with open(CPU_BASELINE_CACHE, "w") as f:
                json.dump({"key": key, "time": time.time(), "baseline": out}, f)
        except OSError:
            pass
        return out
    try:
        with open(CPU_BASELINE_CACHE) as f:
            cached = json.load(f)
        age = time.time() - float(cached.get("time", 0))
        if cached.get("key") == key and 0 <= age <= CPU_BASELINE_MAX_AGE_S:
            out = cached["baseline"]
            out["source"] = ("the N = 1 run's measurement on this box %.0f s earlier (%s; host %s, commit %s)"
                             % (age, CPU_BASELINE_CACHE, key["host"], key["commit"]))
            return out
    except (OSError, ValueError, KeyError, TypeError):
        pass
    out = cpu_baseline(args, model_state)
    out["source"] = "measured on rank 0 of this N = %d run (no matching N = 1 measurement cached on this box)" % world
    return out


def check_one_device_per_rank(group, world, device, backend, shared_gpu):
    """RCCL needs every rank on its own GPU: a launcher that hands two ranks the same device
    (LOCAL_RANK ignored, a too narrow HIP_VISIBLE_DEVICES) deadlocks or silently halves the
    job.  Asserted before the first collective of the data path."""
    import torch.distributed as dist
    visible = torch.cuda.device_count()
    if backend != "nccl" or shared_gpu:
        return {"visible_devices": visible, "distinct_devices": None}
    if world > visible:
        raise SystemExit("bench.py: %d ranks but only %d GPU(s) visible to rank %d"
                         % (world, visible, dist.get_rank(group)))
    # the device's PCI address (domain : bus : device) and uuid identify the physical GPU whatever
    # the visibility masks are; the index inside this process's view separates unmasked ranks
    props = torch.cuda.get_device_properties(device)
    hardware = tuple(str(getattr(props, key, "")) for key in ("pci_domain_id", "pci_bus_id", "pci_device_id", "uuid"))
    known = any(v not in ("", "None") for v in hardware)
    ident = ":".join(hardware) + "/%d/%d" % (visible, device.index)
    idents = [None] * world
    dist.all_gather_object(idents, (ident, known), group=group)
    if not all(k for _, k in idents):       # (no hardware identity on this build of torch: cannot tell)
        return {"visible_devices": visible, "distinct_devices": None}
    if len({i for i, _ in idents}) != world:
        raise SystemExit("bench.py: ranks share a GPU under the nccl backend: %s" % ([i for i, _ in idents],))
    return {"visible_devices": visible, "distinct_devices": world}


class KernelTimer:
    """HIP events around the three MLP entry points, on the stream they are launched on."""

    def __init__(self):
        from fourier_feature_nets_amd import _lib as lib_mod
        self.lib = lib_mod
        self.orig = lib_mod.call
        self.spans = {}
        self.on = False
        lib_mod.call = self._call

    def _call(self, name, *a):
        key = KERNEL_OF.get(name)
        if key is None or not self.on:
            return self.orig(name, *a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.orig(name, *a)
        e1.record()
        self.spans.setdefault(key, []).append((e0, e1))

    def close(self):
        self.lib.call = self.orig

    def summary(self, prog, n_samples):
        specs = prog.layers
        fwd = 2 * sum(sp.out * sp.ld for sp in specs)
        flops = {"mlp_forward_kernel<train>": fwd, "wgrad_unit_kernel": fwd,
                 "mlp_backward_data_kernel": 2 * sum(sp.out * sp.act_in for sp in specs)}
        out = {}
        for key, pairs in self.spans.items():
            ms = [a.elapsed_time(b) for a, b in pairs]
            avg = sum(ms) / len(ms)
            tf = flops[key] * n_samples / (avg * 1e-3) / 1e12
            out[key] = {"avg_ms": round(avg, 4), "launches": len(ms), "achieved": round(tf, 2),
                        "frac": round(tf / F32_MFMA_PEAK_TFLOPS, 4), "flop_per_sample": flops[key]}
        return out


def git_head():
    """Short commit id of this tree: from git, or -- on a GPU box, where .git does not travel --
    from the `.git_head` file scripts/gpu/stamp.sh writes before a gpurun call."""
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True,
                              text=True, timeout=10).stdout.strip()
        if head:
            return head
    except Exception:
        pass
    try:
        with open(os.path.join(ROOT, ".git_head")) as f:
            return f.read().strip() or None
    except OSError:
        return None


def sustained_matrix_rates():
    """The committed record of scripts/probes/mfma_sustained_probe + chain_stream_probe `sustained`
    (power-limited bf16 matrix rates of the whole chip on random three-part operands); None if absent."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), SUSTAINED_PROFILE)
    try:
        with open(path) as f:
            rec = json.load(f)
        return {"file": SUSTAINED_PROFILE,
                "matrix_rate_tflops": rec["reading"]["sustained_matrix_rate_tflops"],
                "operand_stream_tflops": rec["reading"]["sustained_operand_stream_tflops"],
                "note": "back-to-back matrix instructions from registers / the matrix waves' bare stream (LDS operand "
                        "reads + weights out of the L2s), ~30 ms launches: 2500 TFLOP/s needs 2.4 GHz, random operands clock 1.86 / 1.65 GHz"}
    except (OSError, KeyError, ValueError):
        return None


def traffic_of(kernel_name, args):
    """HBM bytes per launch of `kernel_name` from the committed PMC passes (rocprofv3 cannot run
    inside this process).  Returns (bytes | None, provenance)."""
    path = os.path.join(ROOT, TRAFFIC_PROFILE)
    source = {"file": TRAFFIC_PROFILE, "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE "
              "in separate passes over this bench.py; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
              "(gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md section HBM)"}
    if not os.path.exists(path):
        fallback = os.path.join(ROOT, "profiles", "r05_hbm_traffic.json")
        if not os.path.exists(fallback):
            return None, None
        path, source["file"] = fallback, "profiles/r05_hbm_traffic.json"
    with open(path) as f:
        doc = json.load(f)
    source["profiled_commit"] = doc.get("commit")
    if doc.get("config") != {"rays": args.rays, "samples": args.samples} or args.model != "tiny":
        return None, source
    for symbol in SYMBOL_OF[kernel_name]:
        if symbol in doc["kernels"]:
            return doc["kernels"][symbol].get("hbm_bytes"), source
    return None, source


def target_shape_leg(device):
    """MFMA utilisation of the fused Fourier-MLP kernels at the shape BASELINE.json's target is
    stated on: full NeRF, one launch of 65 536 rays x 128 samples (synthetic positions / view
    directions / d_logits; kernels only, HIP events on the launch stream)."""
    import fourier_feature_nets_amd as ffn
    rays, samples = 65536, 128
    n = rays * samples
    torch.cuda.empty_cache()
    torch.manual_seed(20080524)
    model = ffn.NeRF(8, 256, 9, 10, 3, 4, [4], True).to(device)
    prog = model.program()
    gen = torch.Generator(device=device).manual_seed(7)
    pos = torch.rand((n, 3), generator=gen, device=device) * 2 - 1
    view = torch.nn.functional.normalize(torch.randn((n, 3), generator=gen, device=device), dim=1)
    d_logits = torch.randn((n, 4), generator=gen, device=device) / n
    saved = torch.empty((prog.saved_floats(n),), dtype=torch.float32, device=device)
    grads = torch.empty((prog.num_grad_floats,), dtype=torch.float32, device=device)
    timer = KernelTimer()
    try:
        for it in range(3):
            if it == 1:
                timer.on = True
            prog.forward(pos, view, saved)
            prog.backward(d_logits, pos, view, saved, grads)
        torch.cuda.synchronize()
    finally:
        timer.close()
    kernels = timer.summary(prog, n)
    total_ms = sum(k["avg_ms"] for k in kernels.values())
    all_flops = sum(k["flop_per_sample"] for k in kernels.values()) * n
    del saved, pos, view, d_logits
    tf = all_flops / (total_ms * 1e-3) / 1e12
    return {"workload": "NeRF(8,256,9,10,3,4,[4],True), one launch of 65536 rays x 128 samples "
                        "(fused Fourier-MLP kernels only)",
            "kernels": kernels, "mlp_ms": round(total_ms, 3), "achieved": round(tf, 2),
            "frac": round(tf / F32_MFMA_PEAK_TFLOPS, 4), "peak": F32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "dtype": "f32"}


def config3_leg(device, cams, images, bounds, rays_per_step=65536, steps=3):
    """BASELINE configs[2] at full size, driver-visible: one complete optimisation step of the
    full NeRF (8x256 trunk, skip, sigma head, bottleneck, view branch) with S = 128 = 64
    stratified uniform + 64 opacity-guided samples per ray, the coarse model (a tiny NeRF)
    evaluated LIVE on the batch's 64 probe points per ray (no CDF table), 65 536 rays; plus
    frames/sec of the same model through the fused render kernel."""
    import fourier_feature_nets_amd as ffn
    torch.cuda.empty_cache()
    torch.manual_seed(20080524)
    fine = ffn.NeRF(8, 256, 9, 10, 3, 4, [4], True).to(device)
    coarse = ffn.PositionalFourierMLP(3, 4, 5.5).to(device)
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        dataset = ffn.ImageDataset("train", images, bounds, cams, 128, True, True, coarse, 4096,
                                   anneal_start=0.2, num_anneal_steps=2000, device=device,
                                   focus_mode="live")
    torch.cuda.synchronize()
    startup_s = time.perf_counter() - t0
    assert dataset.sampler.cdfs is None
    engine = ffn.TrainEngine(fine, 0.0, None)
    valid_ids = torch.nonzero(dataset.sampler.valid != 0).flatten()
    gen = torch.Generator(device=device).manual_seed(4321)
    prog = fine.program()
    coarse_events = []
    sampler = dataset.sampler
    live_rows = sampler.sample_t
    timer = KernelTimer()

    def timed_rows(index, step):
        # t-sampling of the step = uniform half + the fused coarse pass (probe points -> coarse
        # model -> CDF -> inverse transform -> merge, one launch): timed as one span
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = live_rows(index, step)
        e1.record()
        if timer.on:
            coarse_events.append((e0, e1))
        return out

    sampler.sample_t = timed_rows
    try:
        def run_step(step):
            pick = torch.randint(0, valid_ids.numel(), (rays_per_step,), device=device, generator=gen)
            return engine.train_step(dataset, valid_ids[pick], step, 5e-4 * 0.1 ** (step / 250000))

        run_step(0)
        torch.cuda.synchronize()
        timer.on = True
        t0 = time.perf_counter()
        for step in range(1, 1 + steps):
            loss = run_step(step)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        timer.on = False
        engine.check_finite()
        # TrainEngine bounds its activation workspace: a 65 536 x 128 batch runs as
        # `launches_per_step` forward / backward launches (2 at the default of 2^22 samples)
        launches_per_step = max(1, len(next(iter(timer.spans.values()))) // steps)
        fine_kernels = timer.summary(prog, rays_per_step * 128 // launches_per_step)
    finally:
        timer.close()
        del sampler.sample_t
    coarse_ms = sum(a.elapsed_time(b) for a, b in coarse_events) / steps
    step_ms = 1e3 * elapsed / steps
    mlp_ms = sum(k["avg_ms"] for k in fine_kernels.values()) * launches_per_step
    flop = sum(k["flop_per_sample"] for k in fine_kernels.values()) * rays_per_step * 128
    side = cams[0].resolution.width
    out = {"workload": "lego_%d-shaped full NeRF train step: NeRF(8,256,9,10,3,4,[4],True), "
                       "S = 128 = 64 stratified uniform + 64 opacity-guided samples, live coarse "
                       "model PositionalFourierMLP(3,4,5.5) on 64 probe points per ray, %d rays/step, "
                       "%d cams x %dx%d, exact-f32 MFMA" % (side, rays_per_step, len(cams), side, side),
           "step_ms": round(step_ms, 2), "rays_per_s": round(rays_per_step / (step_ms * 1e-3), 1),
           "steps": steps, "final_loss": float(loss), "sampler_startup_s": round(startup_s, 3),
           "cdf_table_bytes": 0, "sampling_incl_coarse_pass_ms": round(coarse_ms, 3),
           "launches_per_step": launches_per_step,
           "activation_workspace_gb": round(4e-9 * (prog.saved_floats(rays_per_step * 128 // launches_per_step)
                                                    + prog.dz_channels * rays_per_step * 128 // launches_per_step), 1),
           "fine_mlp_ms": round(mlp_ms, 3), "kernels": fine_kernels,
           "fine_mlp_frac_of_f32_mfma_peak": round(flop / (mlp_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
           "whole_step_frac_of_f32_mfma_peak": round(flop / (step_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)}
    # the same step in the OPT-IN split-bf16 mode (labelled; see bf16_train_leg): training kernels
    # of the fine model, and the coarse model's probe pass through the split-bf16 inference kernel
    # (five launches instead of the fused exact-f32 coarse-pass kernel)
    fine.train_precision = "bf16x3"
    coarse.precision = "bf16x3"
    try:
        run_step(steps + 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for step in range(steps + 2, 2 * steps + 2):
            run_step(step)
        torch.cuda.synchronize()
        fast_ms = 1e3 * (time.perf_counter() - t0) / steps
        engine.check_finite()
        out["split_bf16_training"] = {
            "label": "opt-in split-bf16 kernels: training kernels of the fine model, inference kernel "
                     "for the coarse model's probe pass (not the exact-f32 parity mode; reported "
                     "separately)",
            "step_ms": round(fast_ms, 2), "rays_per_s": round(rays_per_step / (fast_ms * 1e-3), 1),
            "speedup_vs_exact_f32_step": round(step_ms / fast_ms, 2)}
    finally:
        fine.train_precision = "f32"
        coarse.precision = "f32"
    # ... and in the OPT-IN f32-accurate split mode (labelled): forward / backward data / weight
    # gradients of the fine model on the three-part kernels (its two 63-channel input windows, the
    # 128-channel view layer and the heads on the exact-f32 units); the coarse pass stays the fused
    # exact-f32 kernel
    fine.train_precision = "bf16x6"
    try:
        run_step(2 * steps + 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for step in range(2 * steps + 3, 3 * steps + 3):
            run_step(step)
        torch.cuda.synchronize()
        acc_ms = 1e3 * (time.perf_counter() - t0) / steps
        engine.check_finite()
        out["f32_accurate_split"] = {
            "label": "opt-in f32-accurate split kernels (three bf16 parts per operand, six products) for "
                     "the fine model's training kernels; fused exact-f32 coarse pass; reported separately",
            "step_ms": round(acc_ms, 2), "rays_per_s": round(rays_per_step / (acc_ms * 1e-3), 1),
            "speedup_vs_exact_f32_step": round(step_ms / acc_ms, 3)}
    finally:
        fine.train_precision = "f32"
    del engine
    prog.release_workspaces()
    torch.cuda.empty_cache()
    # frames/sec of the full NeRF, 128 samples/ray, fused render (a plain uniform sampler like
    # orbit_video without an opacity model: every pixel ray of the frame)
    caster = ffn.Raycaster(fine)
    with contextlib.redirect_stdout(io.StringIO()):
        orbit = ffn.RaySampler(bounds, cams[:2], 128, device=device)
    caster.render_image_device(orbit, 0, 32768)
    torch.cuda.synchronize()
    r0 = time.perf_counter()
    for f in range(2):
        caster.render_image_device(orbit, f, 32768)
    torch.cuda.synchronize()
    out["render_fps_%dx%d_128_samples" % (side, side)] = round(2 / (time.perf_counter() - r0), 3)
    caster.check_finite()
    return out


def config4_leg(device, bounds, cameras=100, size=800):
    """BASELINE configs[3] on one GPU (the 8-GPU run is the driver's): the config-3 step on
    800x800 frames -- 64 M rays of sampler state (2.1 GB), no CDF table (the reference's would be
    16 GB), the same 65 536-ray optimisation step, and the 800x800 frame rate."""
    import fourier_feature_nets_amd as ffn
    torch.cuda.empty_cache()
    intr, poses = synthetic_rig(cameras, size)
    cams = [ffn.CameraInfo.create("train%03d" % i, ffn.Resolution(size, size), intr, p)
            for i, p in enumerate(poses)]
    with contextlib.redirect_stdout(io.StringIO()):
        probe = ffn.RaySampler(bounds, cams, 128, device=device)
        images = analytic_images(probe)
        del probe
    torch.cuda.empty_cache()
    return config3_leg(device, cams, images, bounds, steps=2)


def skip_leg(device, dataset, bounds, rays_per_step, steps=4):
    """OPT-IN empty-space skipping, separately labelled (new semantics, PSNR-level parity; the
    reference evaluates every sample): the tiny-NeRF optimisation step and the fused 400x400
    render with an occupancy grid of the analytic scene (sphere r = 0.6 in the [-1,1]^3 box, 128^3
    cells, dilated) -- the MLP runs only on the samples in occupied cells."""
    import fourier_feature_nets_amd as ffn
    torch.manual_seed(20080524)
    model = ffn.PositionalFourierMLP(3, 4, 5.5).to(device)
    res = 128
    centres = ffn.OccupancyGrid.cell_centres(bounds, res, device)
    logits = torch.zeros((centres.shape[0], 4), device=device)
    logits[:, 3] = torch.where(centres.norm(dim=1) < 0.6, 10.0, -30.0)
    grid = ffn.OccupancyGrid.from_logits(logits, bounds, res, 0.01, True)
    del centres, logits
    engine = ffn.TrainEngine(model, 0.0, None)
    engine.occupancy = grid
    valid_ids = torch.nonzero(dataset.sampler.valid != 0).flatten()
    gen = torch.Generator(device=device).manual_seed(99)

    def run_step(step):
        pick = torch.randint(0, valid_ids.numel(), (rays_per_step,), device=device, generator=gen)
        return engine.train_step(dataset, valid_ids[pick], step, 5e-4)

    run_step(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for step in range(1, 1 + steps):
        run_step(step)
    torch.cuda.synchronize()
    step_ms = 1e3 * (time.perf_counter() - t0) / steps
    engine.check_finite()
    frac = engine.last_evaluated_fraction
    # both opt-in modes together: the occupied samples through the split-bf16 training kernels
    model.train_precision = "bf16x3"
    run_step(1 + steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for step in range(2 + steps, 2 + 2 * steps):
        run_step(step)
    torch.cuda.synchronize()
    both_ms = 1e3 * (time.perf_counter() - t0) / steps
    engine.check_finite()
    model.train_precision = "f32"
    del engine
    caster = ffn.Raycaster(model)
    caster.occupancy = grid
    caster.render_image_device(dataset.sampler, 0, 32768)
    torch.cuda.synchronize()
    r0 = time.perf_counter()
    for f in range(8):
        caster.render_image_device(dataset.sampler, f, 32768)
    torch.cuda.synchronize()
    fps = 8 / (time.perf_counter() - r0)
    caster.check_finite()
    return {"label": "opt-in empty-space skipping: NOT the reference's semantics (samples in empty "
                     "cells are sigma = 0 constants); reported separately from the headline",
            "grid": "128^3 bits, %.1f %% of the cells occupied" % (100 * grid.fraction_occupied()),
            "train_step_ms": round(step_ms, 3),
            "train_rays_per_s": round(rays_per_step / (step_ms * 1e-3), 1),
            "train_step_ms_with_split_bf16_kernels": round(both_ms, 3),
            "train_rays_per_s_with_split_bf16_kernels": round(rays_per_step / (both_ms * 1e-3), 1),
            "evaluated_sample_fraction": round(frac, 4),
            "render_fps_kernels_only": round(fps, 2)}


def config5_leg(device, bounds, rays_per_step=32768, samples=128, cameras=25, size=800, steps=3):
    """BASELINE configs[4] on one GPU: Gaussian Fourier features (sigma = 10), 512-wide MLP
    (the two-waves-per-block "wide" kernels), 800x800 frames, 128 samples/ray -- one optimisation
    step without and with the opt-in occupancy grid (the octree-accelerated skip of the config;
    new semantics, labelled)."""
    import fourier_feature_nets_amd as ffn
    torch.cuda.empty_cache()
    torch.manual_seed(20080524)
    model = ffn.GaussianFourierMLP(3, 4, 10.0, num_channels=512).to(device)
    intr, poses = synthetic_rig(cameras, size)
    cams = [ffn.CameraInfo.create("train%03d" % i, ffn.Resolution(size, size), intr, p)
            for i, p in enumerate(poses)]
    with contextlib.redirect_stdout(io.StringIO()):
        probe = ffn.RaySampler(bounds, cams, samples, device=device)
        images = analytic_images(probe)
        del probe
        dataset = ffn.ImageDataset("train", images, bounds, cams, samples, True, True,
                                   anneal_start=0.2, num_anneal_steps=2000, device=device)
    del images
    valid_ids = torch.nonzero(dataset.sampler.valid != 0).flatten()
    gen = torch.Generator(device=device).manual_seed(555)
    prog = model.program()
    res = 128
    centres = ffn.OccupancyGrid.cell_centres(bounds, res, device)
    logits = torch.zeros((centres.shape[0], 4), device=device)
    logits[:, 3] = torch.where(centres.norm(dim=1) < 0.6, 10.0, -30.0)
    grid = ffn.OccupancyGrid.from_logits(logits, bounds, res, 0.01, True)
    del centres, logits
    out = {"workload": "trex_800-shaped 512-wide Gaussian-feature train step: GaussianFourierMLP(3,4,"
                       "10.0,num_channels=512), %d cams x %dx%d, %d samples/ray, %d rays/step, "
                       "exact-f32 MFMA (wide kernels)" % (cameras, size, size, samples, rays_per_step)}
    for label, occ in (("full", None), ("with_occupancy_grid", grid), ("split_bf16_training", None)):
        # (third entry: the OPT-IN split-bf16 training kernels on the same 512-wide model --
        # two-waves-per-SIMD chain kernels with two output tiles per wave, mlp_bf16_ws.hip, and
        # the split-bf16 weight-gradient units, 256 x 256 windows of the 512-wide layers)
        model.train_precision = "bf16x3" if label == "split_bf16_training" else "f32"
        engine = ffn.TrainEngine(model, 0.0, None)
        engine.occupancy = occ
        timer = KernelTimer()
        try:
            def run_step(step):
                pick = torch.randint(0, valid_ids.numel(), (rays_per_step,), device=device, generator=gen)
                return engine.train_step(dataset, valid_ids[pick], step, 5e-4)

            run_step(0)
            torch.cuda.synchronize()
            timer.on = label == "full"
            t0 = time.perf_counter()
            for step in range(1, 1 + steps):
                run_step(step)
            torch.cuda.synchronize()
            step_ms = 1e3 * (time.perf_counter() - t0) / steps
            engine.check_finite()
        finally:
            timer.close()
        entry = {"step_ms": round(step_ms, 3), "rays_per_s": round(rays_per_step / (step_ms * 1e-3), 1)}
        if label == "split_bf16_training":
            entry["label"] = ("opt-in split-bf16 forward / backward-data / weight-gradient kernels (3 bf16 matrix products "
                              "per f32 product): not the exact-f32 parity mode")
            entry["speedup_vs_exact_f32"] = round(out["full"]["step_ms"] / step_ms, 3)
            model.train_precision = "f32"
        elif occ is None:
            entry["kernels"] = timer.summary(prog, rays_per_step * samples)
        else:
            entry["evaluated_sample_fraction"] = round(engine.last_evaluated_fraction, 4)
            entry["label"] = "opt-in empty-space skipping: not the reference's semantics"
        out[label] = entry
        del engine
        prog.release_workspaces()
        torch.cuda.empty_cache()
    # one 800x800 frame of the same model: the fused kernel's pair-of-waves variant against the
    # three-pass path (sample / model / composite over HBM), without and with the grid
    caster = ffn.Raycaster(model)
    with contextlib.redirect_stdout(io.StringIO()):
        orbit = ffn.RaySampler(bounds, cams[:2], samples, device=device)
    fwd_flop = 2 * sum(sp.out * sp.ld for sp in prog.layers)
    frame_rays = float(orbit.valid.view(2, -1).sum(1, dtype=torch.int64).double().mean().item())
    render = {"rays_per_frame": frame_rays}
    for label, fused, occ in (("fused", "always", None), ("three_pass", False, None),
                              ("fused_with_occupancy_grid", "always", grid),
                              ("three_pass_with_occupancy_grid", False, grid)):
        caster.fused_render, caster.occupancy = fused, occ
        caster.render_image_device(orbit, 0, 32768)
        torch.cuda.synchronize()
        r0 = time.perf_counter()
        for f in range(2):
            caster.render_image_device(orbit, f, 32768)
        torch.cuda.synchronize()
        fps = 2 / (time.perf_counter() - r0)
        render[label + "_fps"] = round(fps, 3)
        if occ is None:
            render[label + "_f32_mfma_frac"] = round(frame_rays * samples * fwd_flop * fps / 1e12
                                                     / F32_MFMA_PEAK_TFLOPS, 4)
    caster.check_finite()
    out["render_%dx%d_%d_samples" % (size, size, samples)] = render
    del caster, orbit
    prog.release_workspaces()
    torch.cuda.empty_cache()
    return out


def bf16_leg(device, bounds, cams, samples):
    """OPT-IN split-bf16 inference mode, separately labelled: every f32 product as three
    v_mfma_f32_32x32x16_bf16 products with f32 accumulation (mlp_bf16.hip).  Frames/sec of the
    400x400 render (three passes: sampling, MLP, compositing) and the error against the exact-f32
    render of the same frames; algorithmic FLOPs against the bf16 matrix peak."""
    import fourier_feature_nets_amd as ffn
    torch.manual_seed(20080524)
    model = ffn.PositionalFourierMLP(3, 4, 5.5).to(device)
    caster = ffn.Raycaster(model)
    with contextlib.redirect_stdout(io.StringIO()):
        sampler = ffn.RaySampler(bounds, cams[:8], samples, False, device=device)
    exact = [caster.render_image_device(sampler, f, 1 << 20).clone() for f in range(2)]
    prog = model.program()
    n = 65536 * samples
    x = torch.rand((n, 3), device=device) * 2 - 1
    flops = 2 * sum(sp.out * sp.ld for sp in prog.layers) * n
    timings = {}
    outs = {}
    for mode, fn in (("f32", lambda: prog.forward(x, None, None)), ("bf16x3", lambda: prog.forward16(x, None))):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            outs[mode] = fn()
        e1.record()
        torch.cuda.synchronize()
        timings[mode] = e0.elapsed_time(e1) / 3
    err = float((outs["f32"] - outs["bf16x3"]).abs().max())
    scale = float(outs["f32"].abs().max())
    del x, outs
    model.precision = "bf16x3"
    fast = [caster.render_image_device(sampler, f, 1 << 20).clone() for f in range(2)]
    torch.cuda.synchronize()
    r0 = time.perf_counter()
    for f in range(8):
        caster.render_image_device(sampler, f, 1 << 20)
    torch.cuda.synchronize()
    fps = 8 / (time.perf_counter() - r0)
    caster.check_finite()
    mse = float(torch.stack([(a.float() - b.float()).square().mean() for a, b in zip(exact, fast)]).mean())
    psnr = 10 * np.log10(255.0 ** 2 / max(mse, 1e-12))
    return {"label": "opt-in split-bf16 inference (3 bf16 matrix products per f32 product, f32 "
                     "accumulation): not the exact-f32 parity mode; reported separately from the headline",
            "mlp_forward_ms": {k: round(v, 3) for k, v in timings.items()},
            "mlp_speedup_vs_exact_f32": round(timings["f32"] / timings["bf16x3"], 2),
            "algorithmic_tflops": round(flops / (timings["bf16x3"] * 1e-3) / 1e12, 1),
            "frac_of_bf16_mfma_peak_2500": round(flops / (timings["bf16x3"] * 1e-3) / 1e12 / 2500.0, 4),
            "matrix_flops_issued_over_algorithmic": 3.0,
            "max_abs_logit_error_vs_f32": err, "max_abs_logit": scale,
            "render_fps_400x400_%d_samples_kernels_only" % samples: round(fps, 2),
            "render_psnr_db_vs_exact_f32_frames": round(float(psnr), 2)}


SPLIT_MODES = {
    "bf16x3": {
        "label": "opt-in split-bf16 training (3 bf16 matrix products per f32 product in the "
                 "forward, backward-data and weight-gradient kernels, f32 accumulation and f32 "
                 "saved activations): not the exact-f32 parity mode; reported separately from the headline",
        "products": 3,
        "kernels": {"ffn_mlp_forward_bf16x3_train": "forward", "ffn_mlp_backward_data_bf16x3": "backward_data",
                    "ffn_mlp_wgrad_units_bf16x3": "weight_gradients"},
        "bound": "forward / backward data (two waves per SIMD, mlp_bf16_ws.hip): matrix pipe busy "
                 "0.43-0.53 at a power-limited 1.8-1.9 GHz, phases separated by workgroup barriers and "
                 "bursts of slab stores; weight gradients: instruction issue and the 39 GB they read "
                 "(DESIGN: split-bf16 sections)"},
    "bf16x6": {
        "label": "opt-in f32-ACCURATE split training (every f32 operand as three bf16 parts = the f32 "
                 "value exactly, six bf16 matrix products per f32 product -- all partial products down "
                 "to 2^-16 of the leading one -- f32 accumulation (backward data: the small products on "
                 "their own accumulator); forward, backward data AND weight gradients on these kernels "
                 "(units with fewer than four 128x128 quadrants stay on the exact-f32 kernel, which folds "
                 "them): error against float64 0.8-0.9x (logits) / 1.0-1.5x (gradients) the "
                 "exact-f32 kernels' own (profiles/r05_bf16x6_probe.json), every reference-golden test "
                 "of the exact mode green in it at the same tolerances (tests/test_round5_gpu.py); "
                 "reported separately: the headline stays the exact-f32 kernels",
        "products": 6,
        "kernels": {"ffn_mlp_forward_bf16x6_train": "forward", "ffn_mlp_backward_data_bf16x6": "backward_data",
                    "ffn_mlp_wgrad_units_bf16x6": "weight_gradients",
                    "ffn_mlp_wgrad_units": "weight_gradients_narrow_units_exact_f32"},
        "bound": "the bf16 matrix pipe at 12 cycles per K (32 for v_mfma_f32_32x32x2_f32).  Forward / backward "
                 "data of the tiny NeRF / Fourier MLP family: matrix waves + vector waves (mlp_bf16_mv.hip: one "
                 "wave per SIMD only multiplies, two generate features and run epilogues; ~38 cycles per matrix "
                 "instruction in the stream, ~45 beside an epilogue: vector instructions beside a saturated "
                 "matrix pipe cost 2-4 cycles each); other chains two waves per SIMD (mlp_bf16_ws.hip, ~0.55 "
                 "busy); weight gradients 192 matrix instructions per block and unit (DESIGN section 4)"},
}


def split_render(device, bounds, cams, samples, mode):
    """Frames/sec of the 400x400 render with the model's inference calls in an opt-in split mode
    (three passes: sampling, MLP, compositing -- the fused render kernel is exact-f32 only), its
    MLP forward next to the exact-f32 one, and the frames' PSNR against the exact-f32 frames."""
    import fourier_feature_nets_amd as ffn
    torch.manual_seed(20080524)
    model = ffn.PositionalFourierMLP(3, 4, 5.5).to(device)
    caster = ffn.Raycaster(model)
    with contextlib.redirect_stdout(io.StringIO()):
        sampler = ffn.RaySampler(bounds, cams[:8], samples, False, device=device)
    exact = [caster.render_image_device(sampler, f, 1 << 20).clone() for f in range(2)]
    prog = model.program()
    n = 65536 * samples
    x = torch.rand((n, 3), device=device) * 2 - 1
    flops = 2 * sum(sp.out * sp.ld for sp in prog.layers) * n
    timings, outs = {}, {}
    for which in ("f32", mode):
        fn = (lambda: prog.forward16(x, None)) if which == "bf16x3" else (lambda: prog.forward(x, None, None, precision=which))
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            outs[which] = fn()
        e1.record()
        torch.cuda.synchronize()
        timings[which] = e0.elapsed_time(e1) / 3
    err = float((outs["f32"] - outs[mode]).abs().max())
    scale = float(outs["f32"].abs().max())
    del x, outs
    model.precision = mode
    fast = [caster.render_image_device(sampler, f, 1 << 20).clone() for f in range(2)]
    torch.cuda.synchronize()
    r0 = time.perf_counter()
    for f in range(8):
        caster.render_image_device(sampler, f, 1 << 20)
    torch.cuda.synchronize()
    fps = 8 / (time.perf_counter() - r0)
    caster.check_finite()
    mse = float(torch.stack([(a.float() - b.float()).square().mean() for a, b in zip(exact, fast)]).mean())
    psnr = 10 * np.log10(255.0 ** 2 / max(mse, 1e-12))
    return {"mlp_forward_ms": {k: round(v, 3) for k, v in timings.items()},
            "mlp_speedup_vs_exact_f32": round(timings["f32"] / timings[mode], 3),
            "algorithmic_tflops": round(flops / (timings[mode] * 1e-3) / 1e12, 1),
            "max_abs_logit_difference_vs_exact_f32_kernels": err, "max_abs_logit": scale,
            "render_fps_400x400_%d_samples_kernels_only" % samples: round(fps, 2),
            "render_psnr_db_vs_exact_f32_frames": round(float(psnr), 2)}


def bf16_train_leg(device, dataset, rays_per_step, samples, steps=6, mode="bf16x3"):
    """An OPT-IN split training mode, separately labelled (`model.train_precision = mode`): the
    tiny-NeRF optimisation step of the headline in that mode next to the exact-f32 step on the same
    box (interleaved: f32, mode, f32, mode), per-kernel times of the mode (HIP events on the launch
    stream), the error of one step's gradients against the exact-f32 kernels on the same batch, and
    how far the losses of the two modes are apart after the same batches."""
    import fourier_feature_nets_amd as ffn
    from fourier_feature_nets_amd import _lib as lib_mod
    info = SPLIT_MODES[mode]
    valid_ids = torch.nonzero(dataset.sampler.valid != 0).flatten()
    out = {"f32": [], mode: []}
    losses, grads, spans = {}, {}, {}
    flop = {}
    orig = lib_mod.call
    recording = [False]

    def hooked(name, *a):
        key = info["kernels"].get(name)
        if key is None or not recording[0]:
            return orig(name, *a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(name, *a)
        e1.record()
        spans.setdefault(key, []).append((e0, e1))

    for which in ("f32", mode, "f32", mode):
        torch.manual_seed(20080524)
        model = ffn.PositionalFourierMLP(3, 4, 5.5).to(device)
        model.train_precision = which
        engine = ffn.TrainEngine(model, 0.0, None)
        gen = torch.Generator(device=device).manual_seed(4321)
        prog = model.program()
        flop = {"forward": 2 * sum(sp.out * sp.ld for sp in prog.layers),
                "backward_data": 2 * sum(sp.out * sp.act_in for sp in prog.layers),
                "weight_gradients": 2 * sum(sp.out * sp.ld for sp in prog.layers)}
        flop["weight_gradients_narrow_units_exact_f32"] = 0

        def run_step(step):
            pick = torch.randint(0, valid_ids.numel(), (rays_per_step,), device=device, generator=gen)
            return engine.train_step(dataset, valid_ids[pick], step, 5e-4)

        run_step(0)
        grads[which] = engine.grads.clone()          # gradients of the first step (identical weights)
        run_step(1)
        torch.cuda.synchronize()
        lib_mod.call = hooked
        recording[0] = which == mode
        t0 = time.perf_counter()
        loss = None
        try:
            for step in range(2, 2 + steps):
                loss = run_step(step)
            torch.cuda.synchronize()
        finally:
            lib_mod.call = orig
        out[which].append(1e3 * (time.perf_counter() - t0) / steps)
        losses[which] = float(loss)
        engine.check_finite()
        del engine, model
        torch.cuda.empty_cache()
    scale = float(grads["f32"].abs().max())
    err = float((grads["f32"] - grads[mode]).abs().max())
    rel_l2 = float((grads["f32"] - grads[mode]).norm() / grads["f32"].norm())
    n_samples = rays_per_step * samples
    kernels = {}
    for key, pairs in spans.items():
        ms = sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)
        tf = flop[key] * n_samples / (ms * 1e-3) / 1e12
        kernels[key] = {"avg_ms": round(ms, 3), "algorithmic_tflops": round(tf, 1),
                        "frac_of_f32_mfma_peak_157.3": round(tf / F32_MFMA_PEAK_TFLOPS, 4)}
        if not key.endswith("exact_f32"):
            kernels[key]["frac_of_bf16_mfma_peak_2500"] = round(tf / 2500.0, 4)
            kernels[key]["matrix_flops_issued_over_algorithmic"] = float(info["products"])
    best = {k: min(v) for k, v in out.items()}
    roofline = None
    if "forward" in kernels:
        # the dominant kernel of the mode against BOTH denominators it is honest against: the f32
        # matrix peak (what the algorithmic FLOP would cost on the exact instruction) and the
        # emulation ceiling of the arithmetic -- the bf16 peak over the products issued per f32 product
        fwd = kernels["forward"]
        ceiling = 2500.0 / info["products"]
        roofline = {"bound": "mfma", "kernel": "%s (training forward)" % [k for k, v in info["kernels"].items() if v == "forward"][0],
                    "achieved": fwd["algorithmic_tflops"], "unit": "TFLOP/s (algorithmic: 2 x MACs of the nn.Linear layers)",
                    "peak": F32_MFMA_PEAK_TFLOPS, "frac": fwd["frac_of_f32_mfma_peak_157.3"],
                    "peak_emulation_ceiling": round(ceiling, 1),
                    "frac_of_emulation_ceiling": round(fwd["algorithmic_tflops"] / ceiling, 4),
                    "emulation_ceiling": "2500 TFLOP/s dense bf16 / %d matrix instructions per f32 product" % info["products"],
                    "algorithmic_flop_per_launch": flop["forward"] * n_samples,
                    "avg_launch_ms": fwd["avg_ms"]}
        sustained = sustained_matrix_rates()
        if mode == "bf16x6" and sustained is not None:
            # a third denominator, MEASURED (committed probe record, another box): what the bf16 matrix pipe
            # sustains under the chip's power budget on operands with these kernels' statistics
            issued = fwd["algorithmic_tflops"] * info["products"]
            roofline["sustained"] = dict(sustained, issued_tflops=round(issued, 1),
                                         frac_of_sustained_matrix_rate=round(issued / sustained["matrix_rate_tflops"], 4),
                                         frac_of_sustained_operand_stream=round(issued / sustained["operand_stream_tflops"], 4))
    organisation = None
    if mode == "bf16x6":
        torch.manual_seed(20080524)
        probe = ffn.PositionalFourierMLP(3, 4, 5.5).to(device).program()
        organisation = {"forward": probe.x6_organisation(), "backward_data": probe.x6_organisation(backward=True),
                        "note": "FFN_BF16X6_ORG=ws keeps the two-waves-per-SIMD kernels (bit-identical slabs, masks, dZ)"}
        del probe
    return {"label": info["label"],
            "chain_kernel_organisation": organisation,
            "roofline": roofline,
            "ms_per_step": round(best[mode], 3),
            "train_step_ms": {k: round(v, 3) for k, v in best.items()},
            "train_step_ms_interleaved_runs": {k: [round(x, 3) for x in v] for k, v in out.items()},
            "train_rays_per_s": round(rays_per_step / (best[mode] * 1e-3), 1),
            "speedup_vs_exact_f32_step": round(best["f32"] / best[mode], 3),
            "kernels": kernels,
            "first_step_gradient_max_abs_error": err, "first_step_gradient_max_abs": scale,
            "first_step_gradient_relative_l2_error": rel_l2,
            "loss_after_%d_steps" % (steps + 2): losses,
            "bound": info["bound"]}


def render_leg(args, caster, sampler, world, rank, barrier):
    """frames/sec of 400x400 renders through the fused kernel: kernels only (frames stay on the
    GPU), with the synchronous D2H copy of each frame (what render_image returns to a caller),
    and with asynchronous copy-out + PNG encoding (FrameSink)."""
    import fourier_feature_nets_amd as ffn
    frames = [f for f in range(8 * world) if f % world == rank]
    caster.render_image(sampler, frames[0], 32768)
    out = {}
    for mode in ("device", "host", "png"):
        torch.cuda.synchronize()
        barrier()
        r0 = time.perf_counter()
        if mode == "device":
            for f in frames:
                caster.render_image_device(sampler, f, 32768)
        elif mode == "host":
            for f in frames:
                caster.render_image(sampler, f, 32768)
        else:
            with tempfile.TemporaryDirectory() as tmp, ffn.FrameSink() as sink:
                for f in frames:
                    sink.submit(caster.render_image_device(sampler, f, 32768),
                                os.path.join(tmp, "frame_%05d.png" % f))
        torch.cuda.synchronize()
        barrier()
        out[mode] = 8 * world / (time.perf_counter() - r0)
    caster.check_finite()
    # rays that hit the volume, per rendered frame (the kernel skips the others after one byte)
    per_cam = sampler.valid.view(sampler.num_cameras, -1).sum(1, dtype=torch.int64)
    out["rays_per_frame"] = float(per_cam[[f % sampler.num_cameras for f in frames]].double().mean().item())
    return out


def render_roofline(prog, render, samples, world):
    """The fused render launch against the f32-MFMA peak: algorithmic FLOP per frame = valid rays
    x samples x forward FLOP per sample (no padding, no skipped rays), at the kernels-only rate
    (per GPU: frames are dealt round-robin to the ranks)."""
    fwd = 2 * sum(sp.out * sp.ld for sp in prog.layers)
    flop_per_frame = render["rays_per_frame"] * samples * fwd
    achieved = flop_per_frame * render["device"] / world / 1e12
    return {"bound": "mfma", "kernel": "render_fused_kernel", "achieved": round(achieved, 2),
            "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4),
            "algorithmic_flop_per_launch": flop_per_frame, "flop_per_sample": fwd}


def default_batch_leg(device, cams, images, bounds, rays=1024, samples=128, steps=600, repeats=5):
    """The reference drivers' DEFAULT batch (train_nerf.py:21-27 / train_tiny_nerf.py: 1024 rays x
    128 samples per step) through TrainEngine.train_step the way `fit` drives it (epoch-level
    validity filter, no per-step host sync), for the tiny and the full NeRF, next to the same
    step at a 32 768-ray batch: the small batch leaves each of the 1024 resident wavefronts
    ~3.1 blocks of 32 samples, which the persistent kernels can only run as 4 rounds.

    Protocol (round 5): `steps` (600) steps per repeat, the MEDIAN of `repeats` (5) repeats; the
    epoch's validity filter (`epoch_ray_ids`: one pass over the epoch's ids with one host
    synchronisation, what `fit` does once per epoch, ray_caster.py:301-305) runs OUTSIDE the timed
    region and is reported on its own; every repeat also records how long the HOST took to enqueue
    its steps (clock stopped before the final synchronize) next to the time the GPU took to run
    them (events around the repeat on the launch stream): enqueue ~ total means the box is
    host-bound and the kernels are not what the number measures."""
    import fourier_feature_nets_amd as ffn
    out = {"workload": "%d rays x %d samples per step (the reference's defaults), 20 cameras 400x400" % (rays, samples),
           "protocol": "%d steps per repeat, median of %d repeats; validity filter of the epoch outside the "
                       "timed region (epoch_filter_ms); host_enqueue = wall time until the last step is "
                       "enqueued, gpu_span = HIP events around the repeat on the launch stream" % (steps, repeats)}
    with contextlib.redirect_stdout(io.StringIO()):
        ds = ffn.ImageDataset("train", images[:20], bounds, cams[:20], samples, True, True, anneal_start=0.2,
                              num_anneal_steps=2000, device=device)
    gen = torch.Generator(device=device).manual_seed(3)
    for name in ("tiny", "nerf"):
        torch.manual_seed(20080524)
        model = (ffn.PositionalFourierMLP(3, 4, 5.5) if name == "tiny"
                 else ffn.NeRF(8, 256, 9, 10, 3, 4, [4], True)).to(device)
        engine = ffn.TrainEngine(model, 0.0, None)

        def measure(batch, count, reps):
            rows = []
            for rep in range(reps + 1):             # (repeat 0 warms up)
                index = torch.randint(0, len(ds), (count * batch,), device=device, generator=gen)
                torch.cuda.synchronize()
                f0 = time.perf_counter()
                ids, cuts = ds.epoch_ray_ids(index, batch)
                torch.cuda.synchronize()
                filter_s = time.perf_counter() - f0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record()
                for i in range(count):
                    engine.train_step(ds, index[i * batch:(i + 1) * batch], 1000 + i, 5e-4,
                                      rays=ids[cuts[i]:cuts[i + 1]])
                e1.record()
                enqueue_s = time.perf_counter() - t0
                torch.cuda.synchronize()
                total_s = time.perf_counter() - t0
                if rep:
                    rows.append({"step_ms": 1e3 * total_s / count, "host_enqueue_ms": 1e3 * enqueue_s / count,
                                 "gpu_span_ms": e0.elapsed_time(e1) / count, "rays": int(ids.numel()) / count,
                                 "epoch_filter_ms": 1e3 * filter_s})
            rows.sort(key=lambda r: r["step_ms"])
            return rows[len(rows) // 2], rows

        def row_of(small, small_all, large):
            per_ray_small = small["step_ms"] / small["rays"]
            per_ray_large = large["step_ms"] / large["rays"]
            return {"ms_per_step": round(small["step_ms"], 4),
                    "ms_per_step_repeats": [round(r["step_ms"], 4) for r in small_all],
                    "host_enqueue_ms_per_step": round(small["host_enqueue_ms"], 4),
                    "gpu_span_ms_per_step": round(small["gpu_span_ms"], 4),
                    "host_bound": bool(small["host_enqueue_ms"] > 0.9 * small["step_ms"]),
                    "epoch_filter_ms": round(small["epoch_filter_ms"], 3),
                    "valid_rays_per_step": round(small["rays"], 1),
                    "rays_per_s": round(1e3 / per_ray_small, 1),
                    "large_batch_ms_per_step": round(large["step_ms"], 3),
                    "large_batch_rays_per_s": round(1e3 / per_ray_large, 1),
                    "per_ray_rate_vs_large_batch": round(per_ray_large / per_ray_small, 4)}

        small, small_all = measure(rays, steps, repeats)
        large, _ = measure(32768, 8, 3)
        engine.check_finite()
        out[name] = row_of(small, small_all, large)
        # the opt-in arithmetic modes at the same batch (separately labelled; three repeats): their
        # passes are 64 (bf16x6) / 128 (bf16x3) samples wide -- a coarser quantisation of the 790-ray
        # batch -- and their shorter kernels leave the host less margin (host_bound says which)
        out[name]["opt_in_modes"] = {}
        for mode in ("bf16x6", "bf16x3"):
            model.train_precision = mode
            small, small_all = measure(rays, steps, 3)
            large, _ = measure(32768, 8, 3)
            engine.check_finite()
            row = row_of(small, small_all, large)
            row["label"] = SPLIT_MODES[mode]["label"]
            row["step_vs_exact_f32_step"] = round(out[name]["ms_per_step"] / row["ms_per_step"], 3)
            out[name]["opt_in_modes"][mode] = row
        model.train_precision = "f32"
        del engine, model
    return out


def spawn_ranks(args):
    """``python bench.py --gpus N`` without a launcher: re-executes itself under
    ``torch.distributed.run`` with N ranks on this node (one per GPU) and passes rank 0's JSON
    line through.  With fewer than N GPUs visible the run is refused unless
    FFN_BENCH_SHARE_GPU=1 (all ranks on cuda:0 over gloo: a functional check of the sharding /
    reduction / timing code on a one-GPU box, never a measurement)."""
    import socket
    visible = torch.cuda.device_count()
    env = dict(os.environ)
    if visible < args.gpus:
        if env.get("FFN_BENCH_SHARE_GPU") != "1":
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (set FFN_BENCH_SHARE_GPU=1 "
                             "for a functional run with all ranks on cuda:0)" % (args.gpus, visible))
        env.setdefault("FFN_BENCH_BACKEND", "gloo")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and there is no CPU fallback")
    # FFN_BENCH_SHARE_GPU=1 (gloo): all ranks on cuda:0 -- a functional check of the N > 1 path
    # on a one-GPU box, not a measurement
    shared_gpu = os.environ.get("FFN_BENCH_SHARE_GPU") == "1"
    if shared_gpu:
        local_rank = 0
        os.environ.setdefault("FFN_BENCH_BACKEND", "gloo")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    group = None
    backend = os.environ.get("FFN_BENCH_BACKEND", "nccl")
    if world > 1 or "RANK" in os.environ:      # under torch.distributed.run: RCCL even for 1 rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        group = dist.group.WORLD
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d (launch with "
                         "torch.distributed.run --nproc-per-node == --gpus, or without a launcher)"
                         % (args.gpus, world))
    placement = None
    if group is not None:
        placement = check_one_device_per_rank(group, world, device, backend, shared_gpu)

    import fourier_feature_nets_amd as ffn

    torch.manual_seed(20080524)
    if args.model == "nerf":
        model = ffn.NeRF(8, 256, 9, 10, 3, 4, [4], True).to(device)
    elif args.model == "gaussian512":
        model = ffn.GaussianFourierMLP(3, 4, 10.0, num_channels=512).to(device)
    else:
        model = ffn.PositionalFourierMLP(3, 4, 5.5).to(device)
    intr, poses = synthetic_rig(args.cameras, args.size)
    cams = [ffn.CameraInfo.create("train%03d" % i, ffn.Resolution(args.size, args.size), intr, p)
            for i, p in enumerate(poses)]
    bounds = np.diag([2, 2, 2, 1]).astype(np.float32)
    with contextlib.redirect_stdout(io.StringIO()):
        probe = ffn.RaySampler(bounds, cams, args.samples, device=device)
        images = analytic_images(probe)
        del probe
        dataset = ffn.ImageDataset("train", images, bounds, cams, args.samples, True, True,
                                   anneal_start=0.2, num_anneal_steps=2000, device=device)
    engine = ffn.TrainEngine(model, 0.0, group)
    if group is not None:
        engine.collective_events = []
    # batches are drawn from the rays that hit the volume, so every step traces exactly
    # rays*world rays (the validity filter of get_rays then keeps all of them)
    valid_ids = torch.nonzero(dataset.sampler.valid != 0).flatten()
    gen = torch.Generator(device=device).manual_seed(1234)
    global_batch, rays_per_gpu = batch_plan(args.rays, world, args.scaling)
    prog = model.program()
    n_samples = rays_per_gpu * args.samples
    timer = KernelTimer()

    def draw():
        pick = torch.randint(0, valid_ids.numel(), (global_batch,), device=device, generator=gen)
        return valid_ids[pick]

    upcoming = {}

    def run_step(step):
        lr = 5e-4 * 0.1 ** (step / 25000)
        if group is None:
            return engine.train_step(dataset, draw(), step, lr)
        # data parallel: the NEXT batch is drawn and filtered now, so that its sampling kernels
        # run under this step's gradient all-reduce (TrainEngine._prefetch)
        batch, rays = upcoming.pop(step, None) or (lambda b: (b, dataset.ray_ids(b)))(draw())
        nxt = draw()
        upcoming[step + 1] = (nxt, dataset.ray_ids(nxt))
        return engine.train_step(dataset, batch, step, lr, rays=rays,
                                 lookahead=(dataset, upcoming[step + 1][1], step + 1))

    def barrier():
        if group is not None:
            import torch.distributed as dist
            dist.barrier(group=group)

    for step in range(args.warmup):
        run_step(step)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    timer.on = True
    if engine.collective_events is not None:
        engine.collective_events.clear()
    t0 = time.perf_counter()
    loss = None
    for step in range(args.warmup, args.warmup + args.steps):
        loss = run_step(step)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.on = False
    timer.close()
    engine.check_finite()
    # (a batch above TrainEngine's launch bound -- 2^22 samples -- runs as several launches)
    launches_per_step = max(1, len(next(iter(timer.spans.values()))) // max(args.steps, 1)) if timer.spans else 1
    n_samples //= launches_per_step
    kernels = timer.summary(prog, n_samples)

    # ---- render leg: 400x400 frames of the first cameras (replicas only: frame f -> rank f%world)
    caster = ffn.Raycaster(model)
    render = None
    if not args.no_render:
        # like orbit_video.py:76-78: a plain (non-stratified) sampler over the frames' cameras
        with contextlib.redirect_stdout(io.StringIO()):
            frame_sampler = ffn.RaySampler(bounds, cams[:8 * world], args.samples, False, device=device)
        render = render_leg(args, caster, frame_sampler, world, rank, barrier)
        del frame_sampler
    rank_ms = None
    if group is not None:
        import torch.distributed as dist
        # every rank's own clock over the timed region; the line's time is the MAX over ranks
        mine = torch.tensor([elapsed], dtype=torch.float64, device=device)
        every = [torch.zeros_like(mine) for _ in range(world)]
        if backend == "nccl":
            dist.all_gather(every, mine, group=group)
        else:
            host = [t.cpu() for t in every]
            dist.all_gather(host, mine.cpu(), group=group)
            every = host
        per_rank = [1e3 * float(t.item()) / args.steps for t in every]
        rank_ms = {"max": max(per_rank), "min": min(per_rank),
                   "slowest_rank": int(np.argmax(per_rank)), "per_rank": [round(v, 4) for v in per_rank]}
        elapsed = max(float(t.item()) for t in every)

    if rank == 0:
        dominant = max(kernels, key=lambda k: kernels[k]["avg_ms"])
        traffic, traffic_source = traffic_of(dominant, args)
        label = {"tiny": ("tiny NeRF", "PositionalFourierMLP(3,4,5.5) 256ch"),
                 "nerf": ("full NeRF", "NeRF(8,256,9,10,3,4,[4],True)"),
                 "gaussian512": ("512-wide Gaussian-feature",
                                 "GaussianFourierMLP(3,4,10.0,num_channels=512)")}[args.model]
        result = {
            "metric": "rays/sec (train)",
            "value": global_batch * args.steps / elapsed,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "antinous_400-shaped %s train step: %d cams x %dx%d, %s, "
                                   "%d samples/ray, %d rays/GPU/step (%d rays per step over %d GPU(s), "
                                   "%s scaling), exact-f32 MFMA"
                                   % (label[0], args.cameras, args.size, args.size, label[1],
                                      args.samples, rays_per_gpu, global_batch, world, args.scaling),
                       "rays_per_gpu": rays_per_gpu, "global_batch_rays": global_batch,
                       "samples_per_ray": args.samples,
                       "parallelism": "dp%d" % world, "final_loss": float(loss),
                       "commit": git_head()},
            "roofline": {"bound": "mfma", "kernel": dominant,
                         "achieved": kernels[dominant]["achieved"],
                         "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": kernels[dominant]["frac"], "traffic": traffic,
                         "traffic_source": traffic_source,
                         "algorithmic_flop_per_launch": kernels[dominant]["flop_per_sample"] * n_samples},
            "kernels": kernels,
            "render": None if render is None else {
                "metric": "frames/sec %dx%d render" % (args.size, args.size),
                "value": render["host"], "frames": 8 * world, "samples_per_ray": args.samples,
                "path": "fused render kernel: one launch per frame (sampling + encoding + MLP + "
                        "compositing + u8 pixels)",
                "kernels_only_fps": render["device"],
                "rays_per_frame": render["rays_per_frame"],
                "roofline": render_roofline(model.program(), render, args.samples, world),
                "with_sync_d2h_fps": render["host"],
                "with_async_d2h_and_png_fps": render["png"],
                "includes": "value = with_sync_d2h_fps: every frame copied to the host before the "
                            "next one starts (what Raycaster.render_image returns)"},
        }
        if engine.collective_events:
            staged = backend != "nccl"
            us = [1e3 * a.elapsed_time(b) for a, _, b in engine.collective_events]
            # (host-staged groups: the middle stamp sits behind the collective, nothing overlaps)
            under = [0.0 if staged else 1e3 * a.elapsed_time(m) for a, m, _ in engine.collective_events]
            waited = ([1e3 * a.elapsed_time(m) for a, m, _ in engine.collective_events] if staged
                      else [1e3 * m.elapsed_time(b) for _, m, b in engine.collective_events])
            step_us = 1e6 * elapsed / args.steps
            result["collective"] = {
                "op": "all_reduce(sum) of [flat gradients | 2 loss sums], one per step",
                "overlap": "issued asynchronously on the communicator's stream; the next step's sampling "
                           "kernels are enqueued under it, the launch stream waits in front of clip+Adam",
                "backend": "rccl" if backend == "nccl" else backend + " (host-staged)",
                "ranks": world, "bytes": int(engine.reduce_buf.numel()) * 4,
                # three stamps on the launch stream per step: issue | look-ahead sampling enqueued |
                # behind the wait.  The span is max(collective, sampling) -- NOT the collective alone
                "span_issue_to_wait_avg_us": round(sum(us) / len(us), 1),
                "span_issue_to_wait_max_us": round(max(us), 1),
                "sampling_under_the_collective_avg_us": round(sum(under) / len(under), 1),
                "launch_stream_waited_avg_us": round(sum(waited) / len(waited), 1),
                "launch_stream_waited_max_us": round(max(waited), 1),
                "waited_frac_of_step": round(sum(waited) / len(waited) / step_us, 4),
                "span_frac_of_step": round(sum(us) / len(us) / step_us, 4),
                "shared_gpu": shared_gpu,
                # rank 0's view of the collective; the ranks' own step times sit next to it
                "rank_step_ms": rank_ms, "placement": placement}
        else:
            result["collective"] = None
        solo = world == 1 and args.model == "tiny"
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        # free the headline leg's buffers before the large reporting-only legs
        del engine, caster
        prog.release_workspaces()
        torch.cuda.empty_cache()
        result["north_star_shape"] = target_shape_leg(device) if solo and not args.no_target_shape else None
        result["config3_step"] = (config3_leg(device, cams, images, bounds)
                                  if solo and not args.no_config3 and args.size == 400 else None)
        result["config4_step"] = (config4_leg(device, bounds)
                                  if solo and not args.no_config3 and args.size == 400 else None)
        result["default_batch_step"] = (default_batch_leg(device, cams, images, bounds)
                                        if solo and not args.no_config3 and args.size == 400 else None)
        result["empty_space_skipping"] = (skip_leg(device, dataset, bounds, args.rays)
                                          if solo and not args.no_skip_leg else None)
        result["split_bf16_inference"] = (bf16_leg(device, bounds, cams, args.samples)
                                          if solo and not args.no_bf16_leg else None)
        result["split_bf16_training"] = (bf16_train_leg(device, dataset, args.rays, args.samples)
                                         if solo and not args.no_bf16_leg else None)
        result["f32_accurate_split"] = (bf16_train_leg(device, dataset, args.rays, args.samples, mode="bf16x6")
                                        if solo and not args.no_bf16_leg else None)
        if result["f32_accurate_split"] is not None:
            result["f32_accurate_split"]["inference"] = split_render(device, bounds, cams, args.samples, "bf16x6")
        del dataset
        torch.cuda.empty_cache()
        result["config5_step"] = (config5_leg(device, bounds)
                                  if solo and not args.no_config5 and args.size == 400 else None)
        result["cpu_baseline"] = (cpu_baseline_for(args, state, world)
                                  if args.model == "tiny" and not args.no_cpu_baseline else None)
        print(json.dumps(result), flush=True)
    if group is not None:
        import torch.distributed as dist
        dist.barrier(group=group)        # rank 0 may still be in its reporting-only legs
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
