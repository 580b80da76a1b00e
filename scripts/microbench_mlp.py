"""Kernel-level micro-benchmark of the fused MLP (forward; backward when available).

    python scripts/microbench_mlp.py [--rays 65536] [--samples 64] [--iters 5]

Prints achieved algorithmic TFLOP/s against the 157.3 TFLOP/s exact-f32 MFMA peak.
"""

import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=65536)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--model", default="positional", choices=["positional", "raw"])
    ap.add_argument("--hidden", type=int, default=3)
    args = ap.parse_args()
    import fourier_feature_nets_amd as ffn
    from fourier_feature_nets_amd.mlp_engine import DenseSpec, EncodingSpec, MlpProgram
    dev = torch.device("cuda:0")
    torch.manual_seed(20080524)
    C = args.channels
    if args.model == "positional":
        b = ffn.PositionalFourierMLP(3, 4, 5.5).b_values.data.clone().to(dev)
        a = torch.ones(b.shape[1], device=dev)
        first = 2 * b.shape[1]
    else:
        b, a, first = None, None, 3
    dims = [(C, first)] + [(C, C)] * (args.hidden - 1) + [(4, C)]
    layers = []
    for i, (o, k) in enumerate(dims):
        lin = torch.nn.Linear(k, o)
        last = i == len(dims) - 1
        layers.append(DenseSpec(lin.weight.detach().to(dev), lin.bias.detach().to(dev),
                                0 if i == 0 else k, 0 if i == 0 else None, not last,
                                (0, 4) if last else None))
    prog = MlpProgram([EncodingSpec(b, a, math.pi, False, dev)], layers, dev)
    prog.pack()
    n = args.rays * args.samples
    x = (torch.rand(n, 3, device=dev) * 2 - 1)
    flops = 2 * sum(o * k for o, k in dims)
    for train in (False, True):
        saved = torch.empty(prog.saved_floats(n), device=dev) if train else None
        prog.forward(x, None, saved)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(args.iters):
            prog.forward(x, None, saved)
        t1.record()
        torch.cuda.synchronize()
        ms = t0.elapsed_time(t1) / args.iters
        tf = flops * n / (ms * 1e-3) / 1e12
        print("fwd%s: n=%d  %.3f ms  %.1f TFLOP/s algorithmic (%.1f%% of 157.3)  %.2f Msamples/s"
              % (" +save" if train else "", n, ms, tf, 100 * tf / 157.3, n / ms / 1e3), flush=True)


if __name__ == "__main__":
    main()
