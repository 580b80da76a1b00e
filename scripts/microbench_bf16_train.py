"""Training kernels of the tiny model at the bench shape, exact-f32 against the opt-in split-bf16
mode: forward (with saved activations), backward (dgrad + wgrad + reduce).
   python scripts/microbench_bf16_train.py [--rays R --samples S]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fourier_feature_nets_amd as ffn  # noqa: E402


def timed(fn, iters=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=65536)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--model", default="tiny")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    if args.model == "tiny":
        model = ffn.PositionalFourierMLP(3, 4, 5.5).to(dev)
        views = None
    else:
        model = ffn.NeRF().to(dev)
        views = torch.nn.functional.normalize(torch.randn(args.rays * args.samples, 3, device=dev), dim=1)
    prog = model.program()
    n = args.rays * args.samples
    x = torch.rand(n, 3, device=dev) * 2 - 1
    saved = torch.empty((prog.saved_floats(n),), dtype=torch.float32, device=dev)
    grads = torch.empty((prog.num_grad_floats,), dtype=torch.float32, device=dev)
    d_logits = torch.randn(n, 4, device=dev) / n
    out = {}
    for mode in ("f32", "bf16x3"):
        out["forward_train_ms " + mode] = round(timed(lambda: prog.forward(x, views, saved, precision=mode)), 3)
        prog.forward(x, views, saved, precision=mode)
        out["backward_ms " + mode] = round(timed(lambda: prog.backward(d_logits, x, views, saved, grads, precision=mode)), 3)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
