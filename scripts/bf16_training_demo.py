"""Quality of the OPT-IN split-bf16 training mode (DESIGN: split-bf16 training): trains the tiny
NeRF on the synthetic sphere scene twice from the same weights and the same batches -- exact-f32
kernels, and `train_precision = "bf16x3"` -- and reports validation PSNR (held-out cameras, full
exact-f32 renders, vs the analytic images) every few hundred steps, the final weights' distance,
and wall time.   python scripts/bf16_training_demo.py [--steps 1200] > profiles/r02_bf16_training_demo.json
"""
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
    ap.add_argument("--steps", type=int, default=1200)
    ap.add_argument("--rays", type=int, default=16384)
    ap.add_argument("--cameras", type=int, default=48)
    ap.add_argument("--size", type=int, default=200)
    ap.add_argument("--report", type=int, default=300)
    ap.add_argument("--model", choices=["tiny", "nerf"], default="tiny")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    intr, poses = B.synthetic_rig(args.cameras + 4, args.size)
    cams = [ffn.CameraInfo.create("c%03d" % i, ffn.Resolution(args.size, args.size), intr, p)
            for i, p in enumerate(poses)]
    bounds = np.diag([2, 2, 2, 1]).astype(np.float32)
    with contextlib.redirect_stdout(io.StringIO()):
        probe = ffn.RaySampler(bounds, cams, 64, device=dev)
        images = B.analytic_images(probe)
        del probe
        held = list(range(0, len(cams), len(cams) // 4))[:4]
        train_ids = [i for i in range(len(cams)) if i not in held]
        train = ffn.ImageDataset("train", images[train_ids], bounds, [cams[i] for i in train_ids], 64, True,
                                 True, anneal_start=0.2, num_anneal_steps=300, device=dev)
        val_sampler = ffn.RaySampler(bounds, [cams[i] for i in held], 64, device=dev)
    val_images = images[held][..., :3].astype(np.float32) / 255
    valid = train.sampler.valid.nonzero().reshape(-1)

    def psnr(model):
        caster = ffn.Raycaster(model)       # exact-f32 fused render for both runs
        mse = 0.0
        for f in range(len(held)):
            frame = caster.render_image(val_sampler, f, 65536).astype(np.float32) / 255
            mse += float(np.mean((frame - val_images[f]) ** 2))
        return -10 * np.log10(mse / len(held))

    runs, weights = {}, {}
    for mode in ("f32", "bf16x3"):
        torch.manual_seed(20080524)
        model = (ffn.PositionalFourierMLP(3, 4, 5.5) if args.model == "tiny"
                 else ffn.NeRF(8, 256, 9, 10, 3, 4, [4], True)).to(dev)
        model.train_precision = mode
        engine = ffn.TrainEngine(model, 0.0, None)
        gen = torch.Generator(device=dev).manual_seed(1)
        torch.cuda.manual_seed(7)            # the stratified jitter stream
        curve, seconds = [], 0.0
        for step in range(args.steps):
            if step % args.report == 0:
                curve.append([step, round(psnr(model), 3)])
                torch.cuda.manual_seed(7 + step)      # renders must not shift the jitter stream
            pick = torch.randint(0, valid.numel(), (args.rays,), generator=gen, device=dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            engine.train_step(train, valid[pick], step, 5e-4 * 0.1 ** (step / 25000))
            torch.cuda.synchronize()
            seconds += time.perf_counter() - t0
        engine.check_finite()
        curve.append([args.steps, round(psnr(model), 3)])
        runs[mode] = {"val_psnr_db": curve, "train_seconds": round(seconds, 2),
                      "ms_per_step": round(1e3 * seconds / args.steps, 3)}
        weights[mode] = torch.cat([p.detach().flatten() for p in model.parameters()])
    rel = float((weights["f32"] - weights["bf16x3"]).norm() / weights["f32"].norm())
    out = {"scene": "synthetic shaded sphere r=0.6, %d train / %d held-out cameras %dx%d, 64 samples/ray, "
                    "%d rays/step, %d steps, %s; same initial weights, batches and jitter"
                    % (len(train_ids), len(held), args.size, args.size, args.rays, args.steps,
                       "tiny NeRF" if args.model == "tiny" else "full NeRF (8x256, skip, view branch)"),
           "note": "opt-in split-bf16 training kernels against the exact-f32 ones; validation frames "
                   "rendered by the exact-f32 fused kernel in both runs",
           "runs": runs,
           "final_psnr_difference_db": round(runs["bf16x3"]["val_psnr_db"][-1][1] - runs["f32"]["val_psnr_db"][-1][1], 4),
           "final_weights_relative_l2_distance": rel}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
