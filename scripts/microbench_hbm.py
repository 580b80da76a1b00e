"""HBM-bound kernels of the path at the bench shape: achieved GB/s of the ALGORITHMIC bytes
 against the 8 TB/s peak.   python scripts/microbench_hbm.py [--rays R --samples S]"""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fourier_feature_nets_amd import ops  # noqa: E402
import fourier_feature_nets_amd as ffn  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=65536)
    ap.add_argument("--samples", type=int, default=64)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    R, S = args.rays, args.samples
    torch.manual_seed(1)
    out = {}
    logits = torch.randn(R, S, 4, device=dev)
    t = torch.sort(torch.rand(R, S, device=dev) * 4 + 2, dim=-1)[0].contiguous()
    sec = timed(lambda: ops.composite_fwd(logits, t, False))
    out["composite_fwd"] = (R * (20 * S + 20), sec)
    d_color = torch.randn(R, 3, device=dev)
    d_alpha = torch.randn(R, device=dev)
    sec = timed(lambda: ops.composite_bwd(logits, t, d_color, d_alpha))
    out["composite_bwd"] = (R * (20 * S + 16 * S + 16), sec)
    # K5t: forward + ground-truth gather / loss sums + backward of a training batch in one launch
    gt_colors = torch.rand(4 * R, 3, device=dev)
    gt_alphas = (torch.rand(4 * R, device=dev) > 0.4).float()
    index = torch.randint(0, 4 * R, (R,), device=dev)
    sec = timed(lambda: ops.composite_train(logits, t, gt_colors, gt_alphas, index, 1.0 / (3 * R), 0.1 / R))
    out["composite_train (K5 + K6 + K5b in one launch)"] = (R * (20 * S + 16 * S + 8 + 16), sec)
    # K2a + K2b in one launch (samplers without an opacity model): t, positions, view directions of a
    # batch -- 28 S B / ray written; read: ray id 8 B + near / far 8 B + start and direction 24 B per
    # ray (+ 4 S B of jitter when stratified)
    from fourier_feature_nets_amd._lib import c_f, c_i, c_i64
    from fourier_feature_nets_amd.ops import _call, _dev
    total = 4 * R
    near_far = torch.stack([torch.full((total,), 3.0, device=dev), torch.full((total,), 5.0, device=dev)]).contiguous()
    starts = torch.randn(total, 3, device=dev)
    dirs = torch.nn.functional.normalize(torch.randn(total, 3, device=dev), dim=1)
    ray_index = torch.randint(0, total, (R,), device=dev)
    unit = torch.linspace(0, 1, S, device=dev)
    noise = torch.rand(R, S, device=dev)
    t_out = torch.empty(R, S, device=dev)
    pos_out = torch.empty(R, S, 3, device=dev)
    view_out = torch.empty(R, S, 3, device=dev)
    for label, jitter in (("uniform", None), ("stratified", noise)):
        sec = timed(lambda: _call("ffn_sample_materialise", _dev(near_far), c_i64(total), _dev(starts), _dev(dirs),
                                  _dev(ray_index, torch.int64, "ray_index"), c_i(R), c_i(S), _dev(unit),
                                  _dev(jitter), c_f(-1.0), _dev(t_out), _dev(pos_out), _dev(view_out)))
        out["sample_materialise (K2a + K2b, %s)" % label] = (R * (28 * S + 40 + (4 * S if jitter is not None else 0)), sec)
    n = R * S
    x = torch.rand(n, 3, device=dev) * 2 - 1
    b = ffn.PositionalFourierMLP(3, 4, 5.5).b_values.data.clone().contiguous().to(dev)
    a = torch.ones(b.shape[1], device=dev)
    sec = timed(lambda: ops.fourier_encode(x, b, a, math.pi, False), iters=5)
    out["fourier_encode (tiny: 2F = 510)"] = (n * (12 + 4 * 2 * b.shape[1]), sec)
    bn = torch.zeros(3, 30)
    for k in range(10):
        for d in range(3):
            bn[d, 3 * k + d] = 2.0 ** k
    bn = bn.to(dev)
    sec = timed(lambda: ops.fourier_encode(x, bn, None, 1.0, True), iters=5)
    out["fourier_encode (NeRF: 63)"] = (n * (12 + 4 * 63), sec)
    # what the memory system gives a pure write / a pure copy of the same size (reference points
    # for the write-dominated encode kernels; not kernels of this library)
    big = torch.empty(n * 2 * b.shape[1], device=dev)
    sec = timed(lambda: big.zero_(), iters=5)
    out["reference: write-only fill of the tiny encode output"] = (big.numel() * 4, sec)
    half = big[: big.numel() // 2]
    other = big[big.numel() // 2:]
    sec = timed(lambda: other.copy_(half), iters=5)
    out["reference: device copy (read + write)"] = (big.numel() * 4, sec)
    res = {k: {"algorithmic_bytes": v[0], "us": round(v[1] * 1e6, 1),
               "GB/s": round(v[0] / v[1] / 1e9, 1), "of_8TB/s": round(v[0] / v[1] / 8e12, 3)}
           for k, v in out.items()}
    print(json.dumps({"rays": R, "samples": S, "kernels": res}, indent=1))


if __name__ == "__main__":
    main()
