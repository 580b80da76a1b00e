"""The split-bf16 chain kernels alone (inference forward, training forward, backward data), the
tiny model at the bench shape and the full NeRF at 2^22 samples.  FFN_BF16_KERNELS=ring selects
the one-wave-per-SIMD ring kernels (A/B within one call).
   python scripts/microbench_bf16_chain.py [--models tiny,nerf]"""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fourier_feature_nets_amd as ffn  # noqa: E402
from fourier_feature_nets_amd._lib import c_i64  # noqa: E402
from fourier_feature_nets_amd.ops import _call, _dev  # noqa: E402


def timed(fn, iters=5):
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
    ap.add_argument("--models", default="tiny,nerf")
    ap.add_argument("--samples", type=int, default=1 << 22)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    out = {"kernels": os.environ.get("FFN_BF16_KERNELS", "ws")}
    for name in args.models.split(","):
        torch.manual_seed(1)
        model = (ffn.PositionalFourierMLP(3, 4, 5.5) if name == "tiny"
                 else ffn.NeRF(8, 256, 9, 10, 3, 4, [4], True)).to(dev)
        prog = model.program()
        n = args.samples
        x = torch.rand(n, 3, device=dev) * 2 - 1
        views = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=1) if model.use_view else None
        saved = torch.empty((prog.saved_floats(n),), dtype=torch.float32, device=dev)
        d_logits = torch.randn(n, 4, device=dev) / n
        ws = prog.workspace(n)
        dz = ws.dz
        _, masks = prog._split_saved(saved, n)
        row = {}
        row["infer_ms"] = round(timed(lambda: prog.forward16(x, views)), 3)
        row["forward_train_ms"] = round(timed(lambda: prog.forward(x, views, saved, precision="bf16x3")), 3)
        row["backward_data_ms"] = round(timed(lambda: _call(
            "ffn_mlp_backward_data_bf16x3", ctypes.byref(prog.bwd16), _dev(prog.packed16_bwd, torch.int16),
            _dev(d_logits), c_i64(n), _dev(masks), _dev(dz))), 3)
        out[name] = row
        del saved, x, views, d_logits
        prog.release_workspaces()
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
