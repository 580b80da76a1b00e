"""Trains the full NeRF (skip trunk + view branch) on the MI355X path (counterpart of the
reference's train_nerf.py: same flags, outputs `nerf.pt` + `log.txt`)."""

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fourier_feature_nets_amd as ffn  # noqa: E402
from scripts import _cli  # noqa: E402


def main():
    args = _cli.build_parser("NeRF training (MI355X)", _cli.TRAIN_COMMON, _cli.NERF_ONLY).parse_args()
    args.device, rank, world, group = _cli.setup_device(args.device, True)
    torch.manual_seed(args.seed)
    model = ffn.NeRF(args.num_layers, args.num_channels, args.pos_max_log_scale, args.pos_freq,
                     args.view_max_log_scale, args.view_freq, [4], not args.omit_inputs)
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
        model.save(os.path.join(args.results_dir, "nerf.pt"))
        _cli.write_log(os.path.join(args.results_dir, "log.txt"), args, log)
    if group is not None:
        torch.distributed.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
