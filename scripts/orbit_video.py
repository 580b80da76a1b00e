"""Renders an orbit of PNG frames from a trained checkpoint on the MI355X path (counterpart of
the reference's orbit_video.py): one fused-kernel launch per frame, copy-out and PNG encoding
overlapped with the next frames (FrameSink).  With several GPUs (torch.distributed.run) frame f
goes to rank f mod world: replicas only, no collective."""

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fourier_feature_nets_amd as ffn  # noqa: E402
from scripts import _cli  # noqa: E402


def main():
    args = _cli.build_parser("Orbit video (MI355X)", _cli.ORBIT).parse_args()
    device, rank, world, _ = _cli.setup_device(args.device, False)
    cameras = ffn.orbit(_cli.axis_vector(args.up_dir), _cli.axis_vector(args.forward_dir),
                        args.num_frames, args.fov_y_degrees,
                        ffn.Resolution(args.resolution, args.resolution), args.distance)
    bounds = np.diag([2, 2, 2, 1]).astype(np.float32)
    model = ffn.load_model(args.model_path)
    if model is None:
        return 1
    model = model.to(device)
    opacity = model
    if args.opacity_model:
        opacity = ffn.load_model(args.opacity_model).to(device)
    mine = list(range(rank, args.num_frames, world))
    caster = ffn.Raycaster(_cli.apply_precision(model, args.precision))
    sampler = ffn.RaySampler(bounds, [cameras[f] for f in mine], args.num_samples, False, opacity,
                             args.batch_size, device=device, focus_mode=_cli.focus_mode(args))
    os.makedirs(args.output_dir, exist_ok=True)
    bar = ffn.ETABar("Rendering", max=len(mine))
    # frames stay on the GPU until the sink's side stream copies them out; PNG encoding runs on
    # worker threads, so the next frame's kernels are already queued while this one is written
    with ffn.FrameSink() as sink:
        for local, frame in enumerate(mine):
            bar.next()
            image = caster.render_image_device(sampler, local, args.batch_size)
            sink.submit(image, os.path.join(args.output_dir, "frame_{:05d}.png".format(frame)))
    caster.check_finite()
    bar.finish()
    return 0


if __name__ == "__main__":
    sys.exit(main())
