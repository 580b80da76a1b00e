"""Shared pieces of the driver scripts: table-driven argument parsers with the reference
drivers' flag names and defaults, log writing, and a small PNG-dumping training hook."""

import argparse
import json
import os

import numpy as np

# not a flag of the reference: how --opacity-model drives the focus samples.  "table" = the
# reference's per-ray CDF table built at start-up (ray_sampler.py:148-166); "live" = no table, the
# coarse model is evaluated per batch inside the sampling kernel -- bit-identical t-values for a
# frozen opacity model, which a checkpoint loaded by these drivers always is, so "auto" means live
FOCUS_MODE = ("--focus-mode", dict(choices=["auto", "table", "live"], default="auto"))

# (flag, kwargs) -- names/defaults follow train_nerf.py:14-71, train_tiny_nerf.py:14-66 and
# orbit_video.py:16-40 of the reference so that command lines carry over unchanged
TRAIN_COMMON = [
    ("data_path", dict(help="dataset NPZ")),
    ("results_dir", dict(help="output directory")),
    ("--mode", dict(choices=["rgba", "rgb", "dilate"], default="rgba")),
    ("--opacity-model", dict(help="checkpoint of a coarse opacity model (focus sampling)")),
    ("--num-samples", dict(type=int, default=128)),
    ("--batch-size", dict(type=int, default=1024)),
    ("--learning-rate", dict(type=float, default=5e-4)),
    ("--num-channels", dict(type=int, default=256)),
    ("--num-steps", dict(type=int, default=50000)),
    ("--report-interval", dict(type=int, default=1000)),
    ("--image-interval", dict(type=int, default=2000)),
    ("--crop-steps", dict(type=int, default=1000)),
    ("--seed", dict(type=int, default=20080524)),
    ("--decay-rate", dict(type=float, default=0.1)),
    ("--weight-decay", dict(type=float, default=0)),
    ("--make-video", dict(action="store_true")),
    ("--color-space", dict(choices=["YCrCb", "RGB"], default="RGB")),
    ("--num-frames", dict(type=int, default=200)),
    ("--device", dict(default="cuda")),
    ("--anneal-start", dict(type=float, default=0.2)),
    ("--num-anneal-steps", dict(type=int, default=2000)),
    # not a flag of the reference: the opt-in split-bf16 kernels, training and renders
    ("--precision", dict(choices=["f32", "bf16x3", "bf16x6"], default="f32")),
    # not flags of the reference either: opt-in empty-space skipping during training (DESIGN K9):
    # exact steps for --skip-warmup steps, then an occupancy grid derived from the model and
    # rebuilt every --skip-refresh steps
    ("--skip-empty-space", dict(action="store_true")),
    ("--skip-warmup", dict(type=int, default=1000)),
    ("--skip-refresh", dict(type=int, default=500)),
    FOCUS_MODE,
]
NERF_ONLY = [
    ("--resolution", dict(type=int, default=400)),
    ("--num-cameras", dict(type=int, default=100)),
    ("--num-layers", dict(type=int, default=8)),
    ("--pos-freq", dict(type=int, default=10)),
    ("--pos-max-log-scale", dict(type=float, default=9)),
    ("--view-freq", dict(type=int, default=4)),
    ("--view-max-log-scale", dict(type=float, default=3)),
    ("--omit-inputs", dict(action="store_true")),
    ("--decay-steps", dict(type=int, default=250000)),
]
TINY_ONLY = [
    ("--embedding-size", dict(type=int, default=256)),
    ("--pos-max-log-scale", dict(type=float, default=5.5)),
    ("--gauss-sigma", dict(type=float, default=6.05)),
    ("--make-activations", dict(action="store_true")),
    ("--decay-steps", dict(type=int, default=25000)),
]
ORBIT = [
    ("model_path", dict(help="trained checkpoint")),
    ("resolution", dict(type=int, help="frame size in pixels")),
    ("output_dir", dict(help="directory for the PNG frames")),
    ("--opacity-model", dict(help="optional checkpoint of an opacity model")),
    ("--distance", dict(type=float, default=4)),
    ("--fov-y-degrees", dict(type=float, default=40)),
    ("--num-frames", dict(type=int, default=200)),
    ("--up-dir", dict(default="y+", choices=["x+", "x-", "y+", "y-", "z+", "z-"])),
    ("--forward-dir", dict(default="z-", choices=["x+", "x-", "y+", "y-", "z+", "z-"])),
    ("--num-samples", dict(type=int, default=128)),
    ("--alpha-thresh", dict(type=float, default=0.3)),
    ("--batch_size", dict(type=int, default=4096)),
    ("--device", dict(default="cuda")),
    ("--precision", dict(choices=["f32", "bf16x3", "bf16x6"], default="f32")),   # not a flag of the reference
    FOCUS_MODE,
]


def build_parser(title, *tables, positional_extra=()):
    parser = argparse.ArgumentParser(title, formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    flat = []
    for table in tables:
        flat.extend(table)
    pos = [row for row in flat if not row[0].startswith("-")]
    opt = [row for row in flat if row[0].startswith("-")]
    for name, kw in pos[:1] + list(positional_extra) + pos[1:] + opt:
        parser.add_argument(name, **kw)
    return parser


def setup_device(requested: str, want_group: bool):
    """Resolves --device for this process and makes it torch's current device (the C ABI
    launches on the current device).  Under ``torch.distributed.run`` (WORLD_SIZE > 1) a plain
    "cuda" becomes cuda:LOCAL_RANK and, if ``want_group``, the RCCL process group is created
    (training: gradients are all-reduced; rendering is replicas-only and needs none).
    Returns (device string, rank, world, group or None)."""
    import torch
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    device = requested
    if world > 1 and device == "cuda":
        device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device(device)
    if dev.type == "cuda":
        torch.cuda.set_device(dev if dev.index is not None else torch.device("cuda", 0))
    group = None
    if want_group and world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = os.environ.get("FFN_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        group = dist.group.WORLD
    return device, rank, world, group


def apply_precision(model, precision: str):
    """--precision bf16x3 / bf16x6: the opt-in split kernels for training and inference calls of a
    fused model (bf16x3: three bf16 products per f32 product, ~2^-16 per product; bf16x6: three-part
    operands, six products, the error of an f32 dot product -- chains of <= 256 channels); the
    default is the exact-f32 kernels.  Not a flag of the reference."""
    if precision != "f32" and hasattr(model, "train_precision"):
        model.precision = precision
        model.train_precision = precision
    return model


def focus_mode(args) -> str:
    """--focus-mode for samplers whose opacity model is a frozen checkpoint: auto -> live."""
    mode = getattr(args, "focus_mode", "auto")
    return "live" if mode == "auto" else mode


def apply_skipping(caster, args):
    """--skip-empty-space: the opt-in occupancy-grid schedule of Raycaster.fit."""
    if getattr(args, "skip_empty_space", False):
        caster.train_occupancy_schedule = (args.skip_warmup, args.skip_refresh)
    return caster


def axis_vector(code):
    vec = np.zeros(3, np.float32)
    vec["xyz".index(code[0])] = 1 if code[1] == "+" else -1
    return vec


def write_log(path, args, log):
    """log.txt in the reference layout: the args as JSON, a blank line, a tab-separated
    header and one row per report."""
    with open(path, "w") as f:
        json.dump(vars(args), f)
        f.write("\n\n")
        f.write("\t".join(["step", "timestamp", "psnr_train", "psnr_val"]) + "\n")
        for e in log:
            f.write("\t".join(str(v) for v in (e.step, e.timestamp, e.train_psnr, e.val_psnr)) + "\n")


def save_png(path, image):
    from PIL import Image
    Image.fromarray(image).save(path)
