"""bf16x6 chain kernels: the matrix-waves / vector-waves organisation (csrc/mlp_bf16_mv.hip) against the
two-waves-per-SIMD one (csrc/mlp_bf16_ws.hip) on the chains both cover -- the forward's buffers compared bit
for bit (slabs, masks), logits to 1e-6; the backward's dZ and gradients to 1e-6 of their largest element with
the zeros of the ReLU masks in the same places (its step 0, the head term, is f32 arithmetic of the vector
waves where the ws kernels issue six bf16 products), and bit for bit against ITSELF across repetitions; then
timings interleaved in one process (FFN_BF16X6_ORG is read per launch).  Run on a GPU box:  python scripts/probes/mv_check.py [--time-only] [--n 4194304]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fourier_feature_nets_amd as ffn  # noqa: E402


def dev():
    return torch.device("cuda:0")


def models():
    torch.manual_seed(7)
    out = {
        "tiny_nerf": ffn.PositionalFourierMLP(3, 4, 5.5, num_layers=3, num_channels=256, embedding_size=256),
        "gaussian": ffn.GaussianFourierMLP(3, 4, 3.0, num_layers=4, num_channels=256, embedding_size=256),
        "positional_8": ffn.PositionalFourierMLP(3, 4, 5.5, num_layers=8, num_channels=256, embedding_size=256),
    }
    return {k: m.to(dev()) for k, m in out.items()}


def run(prog, x, n, org, train):
    os.environ["FFN_BF16X6_ORG"] = org
    buf = torch.zeros((prog.saved_floats(n),), dtype=torch.float32, device=dev()) if train else None
    logits = prog.forward(x, None, buf, precision="bf16x6")
    torch.cuda.synchronize()
    return logits, buf


def close(ref, new, what, tol=1e-6):
    """largest difference over the largest element (the backward's head term is computed in another arithmetic)"""
    scale = max(float(ref.abs().max()), 1e-30)
    err = float((new - ref).abs().max()) / scale
    assert err <= tol, what + (err,)
    return err


def check(name, model):
    prog = model.program()
    worst = worst_b = 0.0
    for n in (1, 31, 64, 65, 1000, 4097, 70000):
        torch.manual_seed(n)
        x = torch.rand(n, 3, device=dev()) * 2 - 1
        ref_l, ref_s = run(prog, x, n, "ws", True)
        new_l, new_s = run(prog, x, n, "mv", True)
        scale = max(float(ref_l.abs().max()), 1.0)
        err = float((new_l - ref_l).abs().max()) / scale
        worst = max(worst, err)
        assert err <= 2e-6, (name, n, "logits", err)
        assert torch.equal(ref_s.view(torch.int32), new_s.view(torch.int32)), (name, n, "slabs / masks differ",
                                                                                int((ref_s.view(torch.int32) != new_s.view(torch.int32)).sum()))
        inf_l, _ = run(prog, x, n, "mv", False)
        assert torch.equal(inf_l, new_l), (name, n, "inference != training forward")
        if prog.bwd_x6 is not None:
            d_logits = torch.randn(n, 4, device=dev()) / n
            ws = prog.workspace(n)
            got = {}
            for org in ("ws", "mv"):
                os.environ["FFN_BF16X6_ORG"] = org
                ws.dz.zero_()
                flat = torch.zeros((prog.num_grad_floats,), dtype=torch.float32, device=dev())
                prog.backward(d_logits, x, None, ref_s, flat, precision="bf16x6")
                torch.cuda.synchronize()
                got[org] = (ws.dz.clone(), flat)
            worst_b = max(worst_b, close(got["ws"][0], got["mv"][0], (name, n, "dZ")), close(got["ws"][1], got["mv"][1], (name, n, "gradients")))
            assert torch.equal(got["ws"][0] == 0, got["mv"][0] == 0), (name, n, "dZ: the masks' zeros moved")
    print("%-14s parity ok (logits within %.1e of the ws kernels', slabs and masks bit-identical; dZ and gradients within %.1e of "
          "their largest element)" % (name, worst, worst_b), flush=True)


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def backward_data(prog, d_logits, buf, n):
    """the backward-DATA kernel alone (prog.backward also runs the weight-gradient units)"""
    import ctypes
    from fourier_feature_nets_amd import mlp_engine as me
    ws = prog.workspace(n)
    _, masks = prog._split_saved(buf, n)
    me._call("ffn_mlp_backward_data_bf16x6", ctypes.byref(prog.bwd_x6), me._dev(prog.packed_x6_bwd, torch.int16),
             me._dev(d_logits), me.c_i64(n), me._dev(masks), me._dev(ws.dz))


def stress(name, model, n, reps):
    """the same launch `reps` times at the bench size: every repetition bit-identical to the ws kernels' buffers
    (a race between the matrix and the vector waves would show as a difference that comes and goes)"""
    prog = model.program()
    torch.manual_seed(2)
    x = torch.rand(n, 3, device=dev()) * 2 - 1
    d_logits = torch.randn(n, 4, device=dev()) / n
    ref_l, ref_s = run(prog, x, n, "ws", True)
    ws = prog.workspace(n)
    os.environ["FFN_BF16X6_ORG"] = "ws"
    backward_data(prog, d_logits, ref_s, n)
    torch.cuda.synchronize()
    ref_dz = ws.dz.clone()
    first_dz = None
    for rep in range(reps):
        new_l, new_s = run(prog, x, n, "mv", True)
        assert torch.equal(ref_s.view(torch.int32), new_s.view(torch.int32)), (name, rep, "slabs / masks differ")
        assert float((new_l - ref_l).abs().max()) <= 2e-7 * max(float(ref_l.abs().max()), 1.0), (name, rep, "logits")
        ws.dz.zero_()
        backward_data(prog, d_logits, new_s, n)
        torch.cuda.synchronize()
        if first_dz is None:
            first_dz = ws.dz.clone()
            close(ref_dz, first_dz, (name, rep, "dZ against the ws kernels'"))
        assert torch.equal(first_dz.view(torch.int32), ws.dz.view(torch.int32)), (name, rep, "dZ differs from the first repetition's")
    print("%-14s %d repetitions at %d samples: slabs / masks bit-identical to the ws kernels' every time, dZ bit-identical to the "
          "first repetition's" % (name, reps, n), flush=True)


def timing(name, model, n, reps):
    prog = model.program()
    torch.manual_seed(1)
    x = torch.rand(n, 3, device=dev()) * 2 - 1
    buf = torch.zeros((prog.saved_floats(n),), dtype=torch.float32, device=dev())
    d_logits = torch.randn(n, 4, device=dev()) / n
    flat = torch.zeros((prog.num_grad_floats,), dtype=torch.float32, device=dev())
    prog.forward(x, None, buf, precision="bf16x6")
    rows = {}
    for rnd in range(2):
        for org in ("ws", "mv"):
            os.environ["FFN_BF16X6_ORG"] = org
            rows.setdefault(org, {}).setdefault("inference_ms", []).append(
                round(timeit(lambda: prog.forward(x, None, None, precision="bf16x6"), reps), 3))
            rows[org].setdefault("train_forward_ms", []).append(
                round(timeit(lambda: prog.forward(x, None, buf, precision="bf16x6"), reps), 3))
            rows[org].setdefault("backward_ms", []).append(
                round(timeit(lambda: prog.backward(d_logits, x, None, buf, flat, precision="bf16x6"), reps), 3))
            rows[org].setdefault("backward_data_ms", []).append(round(timeit(lambda: backward_data(prog, d_logits, buf, n), reps), 3))
    print(json.dumps({"model": name, "samples": n, "ms": rows}), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--time-only", action="store_true")
    ap.add_argument("--check-only", action="store_true")
    ap.add_argument("--stress", type=int, default=0)
    ap.add_argument("--n", type=int, default=4194304)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    ms = models()
    if args.stress:
        for k, m in ms.items():
            stress(k, m, args.n, args.stress)
        sys.exit(0)
    if not args.time_only:
        for k, m in ms.items():
            check(k, m)
    if not args.check_only:
        for k, m in ms.items():
            timing(k, m, args.n, args.reps)
