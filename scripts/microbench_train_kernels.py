"""Per-kernel timing (HIP events around the C-ABI entry points, on the launch stream) of the
training kernels of one model at one launch size, exact-f32 and split-bf16:
   python scripts/microbench_train_kernels.py [--rays R --samples S --model tiny|nerf --iters K]
Set FFN_HIP_LIBRARY to time an alternative build of the library (scripts/probes/make_variants.py)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fourier_feature_nets_amd as ffn  # noqa: E402
from fourier_feature_nets_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=65536)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--model", default="tiny")
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--modes", default="f32,bf16x3")
    ap.add_argument("--channels", type=int, default=256, help="hidden width (nerf / mlp models)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    n = args.rays * args.samples
    if args.model == "tiny":
        model, views = ffn.PositionalFourierMLP(3, 4, 5.5).to(dev), None
    elif args.model == "mlp":
        model, views = ffn.PositionalFourierMLP(3, 4, 5.5, num_channels=args.channels).to(dev), None
    else:
        model = ffn.NeRF(8, args.channels, 9, 10, 3, 4, [4], True).to(dev)
        views = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=1)
    prog = model.program()
    x = torch.rand(n, 3, device=dev) * 2 - 1
    saved = torch.empty((prog.saved_floats(n),), dtype=torch.float32, device=dev)
    grads = torch.empty((prog.num_grad_floats,), dtype=torch.float32, device=dev)
    d_logits = torch.randn(n, 4, device=dev) / n
    spans = {}
    orig = _lib.call
    recording = [False]

    def hooked(name, *a):
        if not recording[0] or not name.startswith("ffn_mlp_") or "pack" in name:
            return orig(name, *a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        orig(name, *a)
        e1.record()
        spans.setdefault(name, []).append((e0, e1))

    _lib.call = hooked
    out = {"library": os.environ.get("FFN_HIP_LIBRARY", "in-tree"), "samples": n, "model": args.model,
           "channels": args.channels,
           "flop_per_sample": {"forward": 2 * sum(sp.out * sp.ld for sp in prog.layers),
                               "backward_data": 2 * sum(sp.out * sp.act_in for sp in prog.layers)}}
    for mode in args.modes.split(","):
        spans.clear()
        for it in range(args.iters + 1):
            recording[0] = it > 0
            prog.forward(x, views, saved, precision=mode)
            prog.backward(d_logits, x, views, saved, grads, precision=mode)
        torch.cuda.synchronize()
        row = {k.replace("ffn_mlp_", ""): round(sum(a.elapsed_time(b) for a, b in v) / len(v), 3) for k, v in spans.items()}
        row["sum_ms"] = round(sum(row.values()), 3)
        # (a digest of the last step's gradients: two builds that claim identical arithmetic can be
        # compared bit for bit across processes)
        import hashlib
        row["grads_sha16"] = hashlib.sha256(grads.cpu().numpy().tobytes()).hexdigest()[:16]
        out[mode] = row
    _lib.call = orig
    print(json.dumps(out))


if __name__ == "__main__":
    main()
