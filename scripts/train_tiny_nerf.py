"""Trains a position-only radiance field on the MI355X path (counterpart of the reference's
train_tiny_nerf.py: same flags, same outputs `tiny_nerf.pt` + `log.txt`)."""

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fourier_feature_nets_amd as ffn  # noqa: E402
from scripts import _cli  # noqa: E402


def main():
    kinds = ("nerf_model", dict(choices=["mlp", "basic", "positional", "gaussian"]))
    args = _cli.build_parser("Tiny NeRF training (MI355X)", _cli.TRAIN_COMMON, _cli.TINY_ONLY,
                             positional_extra=[kinds]).parse_args()
    args.device, rank, world, group = _cli.setup_device(args.device, True)
    torch.manual_seed(args.seed)
    width = dict(num_channels=args.num_channels)
    makers = {
        "mlp": lambda: ffn.MLP(3, 4, **width),
        "basic": lambda: ffn.BasicFourierMLP(3, 4, **width),
        "positional": lambda: ffn.PositionalFourierMLP(3, 4, max_log_scale=args.pos_max_log_scale,
                                                       embedding_size=args.embedding_size, **width),
        "gaussian": lambda: ffn.GaussianFourierMLP(3, 4, sigma=args.gauss_sigma,
                                                   embedding_size=args.embedding_size, **width),
    }
    model = makers[args.nerf_model]()
    opacity = None
    if args.opacity_model:
        opacity = ffn.load_model(args.opacity_model)
        if opacity is None:
            return 1
        opacity = opacity.to(args.device)
    with_alpha = args.mode == "rgba"
    train = ffn.ImageDataset.load(args.data_path, "train", args.num_samples, with_alpha, True,
                                  opacity, args.batch_size, args.color_space,
                                  anneal_start=args.anneal_start,
                                  num_anneal_steps=args.num_anneal_steps, device=args.device,
                                  focus_mode=_cli.focus_mode(args))
    val = ffn.ImageDataset.load(args.data_path, "val", args.num_samples, with_alpha, False,
                                opacity, args.batch_size, args.color_space, device=args.device,
                                focus_mode=_cli.focus_mode(args))
    if train is None or val is None:
        return 1
    if args.mode == "dilate":
        train.mode = ffn.RayDataset.Mode.Dilate
    os.makedirs(args.results_dir, exist_ok=True)
    if args.make_activations and rank == 0:
        # (train_tiny_nerf.py:137-146 of the reference adds an ActivationVisualizer: a lecture
        # visualisation of per-layer activations, outside the HIP hot path -- SURVEY section 2)
        print("warning: --make-activations is not supported on the HIP path (the fused kernels "
              "keep hidden activations on the CU); training continues without the activation "
              "video", file=sys.stderr)
    caster = _cli.apply_skipping(ffn.Raycaster(_cli.apply_precision(model.to(args.device), args.precision)), args)
    caster.process_group = group      # data parallel under torch.distributed.run
    if world > 1:                     # distinct jitter streams; weights are broadcast by fit
        torch.cuda.manual_seed(args.seed + rank)
    if rank != 0:
        hooks = []
    elif args.make_video:      # same choice of visualizers as the reference driver
        hooks = [ffn.OrbitVideoVisualizer(args.results_dir, args.num_steps,
                                          train.cameras[0].resolution, args.num_frames,
                                          args.num_samples, args.color_space, device=args.device)]
    else:
        hooks = [ffn.EvaluationVisualizer(args.results_dir, ds, args.image_interval)
                 for ds in (train, val)]
    log = caster.fit(train, val, args.batch_size, args.learning_rate, args.num_steps,
                     args.crop_steps, args.report_interval, args.decay_rate, args.decay_steps,
                     args.weight_decay, hooks)
    if rank == 0:
        model.save(os.path.join(args.results_dir, "tiny_nerf.pt"))
        _cli.write_log(os.path.join(args.results_dir, "log.txt"), args, log)
    if group is not None:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
