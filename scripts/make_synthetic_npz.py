"""Writes a procedural RGBA dataset in the reference's NPZ schema (images, intrinsics,
extrinsics, bounds, split_counts) -- the real assets cannot be downloaded here.

    python scripts/make_synthetic_npz.py out.npz --size 400 --cameras 120
"""

import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--size", type=int, default=400)
    ap.add_argument("--cameras", type=int, default=120)
    args = ap.parse_args()
    import contextlib
    import io
    import fourier_feature_nets_amd as ffn
    from bench import analytic_images, synthetic_rig
    intr, poses = synthetic_rig(args.cameras, args.size)
    cams = [ffn.CameraInfo.create("c%03d" % i, ffn.Resolution(args.size, args.size), intr, p)
            for i, p in enumerate(poses)]
    bounds = np.diag([2, 2, 2, 1]).astype(np.float32)
    with contextlib.redirect_stdout(io.StringIO()):
        sampler = ffn.RaySampler(bounds, cams, 8)
    images = analytic_images(sampler)
    n_val = max(1, args.cameras // 17)
    n_test = max(1, args.cameras // 9)
    order = torch.randperm(args.cameras, generator=torch.Generator().manual_seed(0)).numpy()
    np.savez(args.path, images=images[order], intrinsics=np.stack([intr] * args.cameras),
             extrinsics=np.stack(poses)[order], bounds=bounds,
             split_counts=np.array([args.cameras - n_val - n_test, n_val, n_test], np.int32))
    print("wrote", args.path, images.shape)


if __name__ == "__main__":
    main()
