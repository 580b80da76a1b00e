"""Benchmark of the NeRF volume-rendering hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" is one complete optimisation step of the reference's training loop
(ray_caster.py:319-329) on one batch of synthetic rays: stratified t-sampling -> fused
Fourier-MLP forward -> alpha compositing -> loss -> backward (composite, dgrad, wgrad) ->
[RCCL all-reduce] -> clip + Adam.  Workload = BASELINE.json configs[1]: tiny NeRF
(PositionalFourierMLP(3,4,5.5), 256 channels) on a synthetic 100 x 400x400 RGBA dataset,
64 samples/ray, 65536 rays per GPU per step (weak scaling).  Inputs (ray state, ground truth,
weights) are resident in HBM before the timed region.

Rank 0 prints ONE JSON line (metric, value, roofline of the dominant kernel, CPU baseline).
"""

import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rays", type=int, default=65536, help="rays per GPU per step")
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--cameras", type=int, default=100)
    ap.add_argument("--size", type=int, default=400)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-render", action="store_true", help="skip the frames/sec leg")
    ap.add_argument("--no-target-shape", action="store_true",
                    help="skip the MFMA-utilisation leg at the north-star shape (NeRF, 65536 x 128)")
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


def cpu_baseline(args, model_state, log):
    """The oracle's training step (the reference's ATen op sequence restated) on the host
    cores, on a bounded sample of the same workload: full training step and forward only."""
    from oracle import ffn_oracle as orc
    # torch's CPU kernels stop scaling (and then regress) far below the core count of a GPU
    # host: `cores` = the threads actually used (the fastest count for this op mix, at most one
    # per physical core); the host's physical / logical counts are reported next to it
    phys, logical = physical_cores()
    cores = min(phys, 32)
    torch.set_num_threads(cores)
    rays, S = 1024, args.samples
    rng = torch.Generator().manual_seed(1)
    ws = [model_state["layers.%d.weight" % i].cpu() for i in range(4)]
    bs = [model_state["layers.%d.bias" % i].cpu() for i in range(4)]
    model = orc.OracleFourierMLP(model_state["a_values"].cpu(), model_state["b_values"].cpu(), ws, bs)
    trainer = orc.OracleTrainer(model, 5e-4)
    near = torch.full((rays,), 3.0)
    far = torch.full((rays,), 5.0)
    starts = torch.randn(rays, 3, generator=rng)
    starts = 4 * starts / starts.norm(dim=-1, keepdim=True)
    dirs = -starts / 4
    gt_c, gt_a = torch.rand(rays, 3, generator=rng), (torch.rand(rays, generator=rng) > 0.4).float()

    def one_step():
        noise = torch.rand((rays, S), generator=rng)
        t = orc.uniform_t(near, far, S, noise)
        pos = starts.unsqueeze(1) + t.unsqueeze(-1) * dirs.unsqueeze(1)
        trainer.step(pos, None, t, gt_c, gt_a, 5e-4)

    one_step()
    t0 = time.time()
    done = 0
    while time.time() - t0 < 10.0 and done < 40 or done < 2:
        one_step()
        done += 1
    elapsed = time.time() - t0
    # forward only (sampling + model + compositing, no autograd): the render-side baseline
    t1 = time.time()
    fwd_done = 0
    with torch.no_grad():
        while time.time() - t1 < 5.0 and fwd_done < 40 or fwd_done < 2:
            noise = torch.rand((rays, S), generator=rng)
            t = orc.uniform_t(near, far, S, noise)
            pos = starts.unsqueeze(1) + t.unsqueeze(-1) * dirs.unsqueeze(1)
            trainer.loss(pos, None, t, gt_c, gt_a)
            fwd_done += 1
    fwd_elapsed = time.time() - t1
    return {"value": rays * done / elapsed, "unit": "rays/s", "cores": cores,
            "physical_cores": phys, "logical_cpus": logical, "kind": "port",
            "forward_only_rays_per_s": rays * fwd_done / fwd_elapsed,
            "sample": "%d training steps (%.1f s) and %d forward passes (%.1f s) of %d rays x %d "
                      "samples (oracle: the reference's ATen op sequence on the host CPU, "
                      "torch.set_num_threads(%d))" % (done, elapsed, fwd_done, fwd_elapsed, rays,
                                                      S, cores)}


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
    from fourier_feature_nets_amd import _lib as lib_mod
    orig_call = lib_mod.call
    spans = {}

    def timed(name, *a):
        key = {"ffn_mlp_forward": "mlp_forward_kernel<train>",
               "ffn_mlp_backward_data": "mlp_backward_data_kernel",
               "ffn_mlp_wgrad_units": "wgrad_unit_kernel"}.get(name)
        if key is None:
            return orig_call(name, *a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig_call(name, *a)
        e1.record()
        spans.setdefault(key, []).append((e0, e1))

    lib_mod.call = timed
    try:
        for it in range(3):
            if it == 1:
                spans.clear()
            prog.forward(pos, view, saved)
            prog.backward(d_logits, pos, view, saved, grads)
        torch.cuda.synchronize()
    finally:
        lib_mod.call = orig_call
    specs = prog.layers
    fwd = 2 * sum(sp.out * sp.ld for sp in specs)
    flops = {"mlp_forward_kernel<train>": fwd, "wgrad_unit_kernel": fwd,
             "mlp_backward_data_kernel": 2 * sum(sp.out * sp.act_in for sp in specs)}
    out = {}
    total_ms = 0.0
    for key, pairs in spans.items():
        ms = sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)
        total_ms += ms
        tf = flops[key] * n / (ms * 1e-3) / 1e12
        out[key] = {"avg_ms": round(ms, 3), "achieved": round(tf, 2),
                    "frac": round(tf / F32_MFMA_PEAK_TFLOPS, 4), "flop_per_sample": flops[key]}
    all_flops = sum(flops.values()) * n
    del saved, pos, view, d_logits
    torch.cuda.empty_cache()
    return {"workload": "NeRF(8,256,9,10,3,4,[4],True), one launch of 65536 rays x 128 samples "
                        "(fused Fourier-MLP kernels only)",
            "kernels": out, "mlp_ms": round(total_ms, 3),
            "achieved": round(all_flops / (total_ms * 1e-3) / 1e12, 2),
            "frac": round(all_flops / (total_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
            "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "dtype": "f32"}


def spawn_ranks(args):
    """``python bench.py --gpus N`` without a launcher: re-executes itself under
    ``torch.distributed.run`` with N ranks on this node (one per GPU) and passes rank 0's JSON
    line through.  With fewer than N GPUs visible the run is refused unless
    FFN_BENCH_SHARE_GPU=1 (all ranks on cuda:0 over gloo: a functional check of the sharding /
    reduction / timing code on a one-GPU box, never a measurement)."""
    import socket
    import subprocess
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
    # FFN_BENCH_SHARE_GPU=1 (with FFN_BENCH_BACKEND=gloo): all ranks on cuda:0 -- a functional
    # check of the N > 1 path on a one-GPU box, not a measurement
    shared_gpu = os.environ.get("FFN_BENCH_SHARE_GPU") == "1"
    if shared_gpu:
        local_rank = 0
        os.environ.setdefault("FFN_BENCH_BACKEND", "gloo")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    group = None
    if world > 1 or "RANK" in os.environ:      # under torch.distributed.run: RCCL even for 1 rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("FFN_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        group = dist.group.WORLD
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d (launch with "
                         "torch.distributed.run --nproc-per-node == --gpus, or without a launcher)"
                         % (args.gpus, world))

    import contextlib
    import io
    import fourier_feature_nets_amd as ffn
    from fourier_feature_nets_amd import ops

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
    quiet = io.StringIO()
    with contextlib.redirect_stdout(quiet):
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
    global_batch = args.rays * world
    prog = model.program()
    n_samples = args.rays * args.samples

    # per-kernel timing with events on the launch stream
    timers = {"fwd": [], "dgrad": [], "wgrad": []}
    from fourier_feature_nets_amd import _lib as lib_mod
    orig_call = lib_mod.call

    def timed_call(name, *a):
        key = {"ffn_mlp_forward": "fwd", "ffn_mlp_backward_data": "dgrad",
               "ffn_mlp_wgrad_units": "wgrad"}.get(name)
        if key is None or not timed_call.on:
            return orig_call(name, *a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig_call(name, *a)
        e1.record()
        timers[key].append((e0, e1))

    timed_call.on = False
    lib_mod.call = timed_call

    def run_step(step):
        pick = torch.randint(0, valid_ids.numel(), (global_batch,), device=device, generator=gen)
        batch = valid_ids[pick]
        lr = 5e-4 * 0.1 ** (step / 25000)
        return engine.train_step(dataset, batch, step, lr)

    def barrier():
        if group is not None:
            import torch.distributed as dist
            dist.barrier(group=group)

    for step in range(args.warmup):
        run_step(step)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    timed_call.on = True
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
    timed_call.on = False
    engine.check_finite()

    # ---- render leg: 400x400 frames of the first cameras (replicas only: frame f -> rank f%world)
    caster = ffn.Raycaster(model)
    frames = [] if args.no_render else [f for f in range(8 * world) if f % world == rank]
    for f in frames[:1]:
        caster.render_image(dataset.sampler, f, 32768)
    torch.cuda.synchronize()
    barrier()
    r0 = time.perf_counter()
    for f in frames:
        caster.render_image(dataset.sampler, f, 32768)
    torch.cuda.synchronize()
    barrier()
    render_s = time.perf_counter() - r0
    if group is not None:
        import torch.distributed as dist
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=group)
        elapsed = float(tmax.item())

    if rank == 0:
        specs = prog.layers
        fwd_flops = 2 * sum(sp.out * sp.ld for sp in specs)
        dgrad_flops = 2 * sum(sp.out * sp.act_in for sp in specs)      # no dgrad into encodings
        flops = {"fwd": fwd_flops, "dgrad": dgrad_flops, "wgrad": fwd_flops}
        kernels = {}
        for key, pairs in timers.items():
            if not pairs:
                continue
            ms = [a.elapsed_time(b) for a, b in pairs]
            avg = sum(ms) / max(len(ms), 1)
            achieved = flops[key] * n_samples / (avg * 1e-3) / 1e12 if avg > 0 else 0.0
            kernels[key] = {"avg_ms": round(avg, 4), "launches": len(ms),
                            "achieved": round(achieved, 2),
                            "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4),
                            "flop_per_sample": flops[key]}
        dominant = max(kernels, key=lambda k: kernels[k]["avg_ms"])
        # HBM bytes per launch from the PMC passes committed under profiles/ (same workload)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        symbol = {"fwd": "ffn::mlp_forward_kernel<1, false>",
                  "dgrad": "ffn::mlp_backward_data_kernel<false>",
                  "wgrad": "ffn::wgrad_unit_kernel"}
        if os.path.exists(tpath):
            with open(tpath) as f:
                tdata = json.load(f)
            if tdata["config"] == {"rays": args.rays, "samples": args.samples} and args.model == "tiny":
                traffic = tdata["kernels"].get(symbol[dominant], {}).get("hbm_bytes")
        names = {"fwd": "mlp_forward_kernel<train>", "dgrad": "mlp_backward_data_kernel",
                 "wgrad": "wgrad_unit_kernel"}
        result = {
            "metric": "rays/sec (train)",
            "value": global_batch * args.steps / elapsed,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "antinous_400-shaped %s train step: %d cams x %dx%d, %s, "
                                   "%d samples/ray, %d rays/GPU/step, exact-f32 MFMA"
                                   % ({"tiny": "tiny NeRF", "nerf": "full NeRF",
                                       "gaussian512": "512-wide Gaussian-feature"}[args.model],
                                      args.cameras, args.size, args.size,
                                      {"tiny": "PositionalFourierMLP(3,4,5.5) 256ch",
                                       "nerf": "NeRF(8,256,9,10,3,4,[4],True)",
                                       "gaussian512": "GaussianFourierMLP(3,4,10.0,num_channels=512)"}[args.model],
                                      args.samples, args.rays),
                       "rays_per_gpu": args.rays, "samples_per_ray": args.samples,
                       "parallelism": "dp%d" % world, "final_loss": float(loss)},
            "roofline": {"bound": "mfma", "kernel": names[dominant],
                         "achieved": kernels[dominant]["achieved"],
                         "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": kernels[dominant]["frac"], "traffic": traffic,
                         "algorithmic_flop_per_launch": kernels[dominant]["flop_per_sample"] * n_samples},
            "kernels": {names[k]: v for k, v in kernels.items()},
            "render": None if args.no_render else {
                       "metric": "frames/sec %dx%d render" % (args.size, args.size),
                       "value": 8 * world / render_s, "frames": 8 * world,
                       "samples_per_ray": args.samples, "includes": "sampling, fused MLP, "
                       "composite, u8 assembly and the D2H copy of each frame"},
        }
        if engine.collective_events:
            us = [1e3 * a.elapsed_time(b) for a, b in engine.collective_events]
            backend = os.environ.get("FFN_BENCH_BACKEND", "nccl")
            result["collective"] = {
                "op": "all_reduce(sum) of [flat gradients | 2 loss sums], one per step",
                "backend": "rccl" if backend == "nccl" else backend + " (host-staged)",
                "ranks": world, "bytes": int(engine.reduce_buf.numel()) * 4,
                "avg_us": round(sum(us) / len(us), 1), "max_us": round(max(us), 1),
                "frac_of_step": round(sum(us) / len(us) * 1e-3 / (1e3 * elapsed / args.steps), 4),
                "shared_gpu": shared_gpu}
        else:
            result["collective"] = None
        result["north_star_shape"] = None if (args.no_target_shape or args.model != "tiny") \
            else target_shape_leg(device)
        if not args.no_cpu_baseline and world == 1 and args.model == "tiny":
            state = {k: v.detach() for k, v in model.state_dict().items()}
            result["cpu_baseline"] = cpu_baseline(args, state, None)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    if group is not None:
        import torch.distributed as dist
        dist.barrier(group=group)        # rank 0 may still be in its reporting-only legs
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
