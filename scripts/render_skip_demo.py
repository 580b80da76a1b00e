"""Empty-space skipping on a (briefly) trained model: frames/sec and PSNR of the skipped render
against the full one.   python scripts/render_skip_demo.py [--steps 300] [--resolution 128]

New behaviour relative to the reference: not part of bench.py's metric."""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
import fourier_feature_nets_amd as ffn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--rays", type=int, default=65536)
    ap.add_argument("--resolution", type=int, default=128)
    ap.add_argument("--threshold", type=float, default=0.01)
    ap.add_argument("--frames", type=int, default=8)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(20080524)
    model = ffn.PositionalFourierMLP(3, 4, 5.5).to(dev)
    intr, poses = B.synthetic_rig(100, 400)
    cams = [ffn.CameraInfo.create("t%03d" % i, ffn.Resolution(400, 400), intr, p) for i, p in enumerate(poses)]
    bounds = np.diag([2, 2, 2, 1]).astype(np.float32)
    with contextlib.redirect_stdout(io.StringIO()):
        probe = ffn.RaySampler(bounds, cams, 64, device=dev)
        images = B.analytic_images(probe)
        del probe
        ds = ffn.ImageDataset("train", images, bounds, cams, 64, True, True, device=dev)
    engine = ffn.TrainEngine(model, 0.0, None)
    valid = ds.sampler.valid.nonzero().reshape(-1)
    gen = torch.Generator(device=dev).manual_seed(1)
    for step in range(args.steps):
        pick = torch.randint(0, valid.numel(), (args.rays,), generator=gen, device=dev)
        loss = engine.train_step(ds, valid[pick], step, 5e-4)
    caster = ffn.Raycaster(model)

    def render_all():
        caster.render_image(ds.sampler, 0, 65536)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        frames = [caster.render_image(ds.sampler, f, 65536) for f in range(args.frames)]
        torch.cuda.synchronize()
        return frames, args.frames / (time.perf_counter() - t0)

    full, fps_full = render_all()
    t0 = time.perf_counter()
    grid = ffn.OccupancyGrid.from_model(model, bounds, args.resolution, args.threshold)
    torch.cuda.synchronize()
    build_ms = (time.perf_counter() - t0) * 1e3
    caster.occupancy = grid
    skipped, fps_skip = render_all()
    psnr = []
    for a, b in zip(full, skipped):
        mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
        psnr.append(10 * np.log10(255.0 ** 2 / max(mse, 1e-12)))
    print(json.dumps({"train_steps": args.steps, "final_loss": float(loss),
                      "grid": "%d^3" % args.resolution, "sigma_threshold": args.threshold,
                      "cells_occupied": round(grid.fraction_occupied(), 4),
                      "grid_build_ms": round(build_ms, 1),
                      "fps_full": round(fps_full, 2), "fps_skipping": round(fps_skip, 2),
                      "psnr_skipped_vs_full_db": [round(p, 2) for p in psnr]}))


if __name__ == "__main__":
    main()
