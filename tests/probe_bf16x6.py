"""The probe that gates the f32-accurate split mode ("bf16x6": three bf16 parts per operand, six --
or all nine -- partial products per f32 product, f32 accumulation; csrc/mlp_bf16_ws.hip):

  (1) ERROR against float64 of the same network: raw-input ReLU MLPs of 256 channels (pure matrix
      arithmetic: no encoding whose f32 angle would dominate every mode alike), logits and every
      weight gradient, for the exact-f32 kernels, bf16x3, bf16x6 with 6 and with 9 products; for
      the tiny NeRF and the full NeRF (whose features all modes but bf16x3 generate bit-identically)
      the distance of each split mode from the exact-f32 kernels next to it;
  (2) TIME of the chain kernels at 2^22 samples (inference forward, training forward, backward
      data), per mode, and what it means per K block of a 256 -> 256 layer.

Stop rule: the mode is built out only if its K loops are >= 1.25x the
exact-f32 ones at an error <= 2x the exact kernels' own.

    python -m tests.probe_bf16x6 --out gpurun_out/r5probe/bf16x6_probe.json

Test infrastructure (float64 torch references live here, not in the product)."""

import argparse
import ctypes
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MODES = [("f32", None), ("bf16x3", None), ("bf16x6", "6"), ("bf16x6", "9")]


def label(mode, products):
    return mode if products is None else "%s_%sp" % (mode, products)


def set_products(products):
    if products is None:
        os.environ.pop("FFN_BF16X6_PRODUCTS", None)
    else:
        os.environ["FFN_BF16X6_PRODUCTS"] = products


def mlp_f64(model, x):
    """float64 forward of a raw-input ``ffn.MLP`` (fourier_feature_models.py:70-78 without the
    encoding)."""
    h = x.double()
    last = len(model.layers) - 1
    params = []
    for i, layer in enumerate(model.layers):
        w = layer.weight.detach().double().requires_grad_(True)
        b = layer.bias.detach().double().requires_grad_(True)
        params += [w, b]
        h = h @ w.t() + b
        if i != last:
            h = torch.relu(h)
    return h, params


def error_table(dev, layers, n=8192, seed=0):
    import fourier_feature_nets_amd as ffn
    torch.manual_seed(seed)
    model = ffn.MLP(3, 4, num_layers=layers, num_channels=256).to(dev)
    x = (torch.rand(n, 3, device=dev) * 2 - 1)
    probe = torch.randn(n, 4, device=dev) / n
    want, params64 = mlp_f64(model, x)
    (want * probe.double()).sum().backward()
    scale = float(want.abs().max())
    rows = {}
    names = [n for i in range(len(model.layers)) for n in ("layers.%d.weight" % i, "layers.%d.bias" % i)]
    for mode, products in MODES:
        set_products(products)
        model.precision = model.train_precision = mode
        with torch.no_grad():
            y_inf = model(x)
        model.zero_grad()
        y = model(x)
        (y * probe).sum().backward()
        err = (y.detach().double() - want.detach())
        row = {"logits_max_abs_err_over_max_abs": float(err.abs().max()) / scale,
               "logits_rms_err_over_rms": float(err.pow(2).mean().sqrt() / want.detach().pow(2).mean().sqrt()),
               "inference_equals_training_forward": bool(torch.equal(y_inf, y.detach()))}
        gmax, grms, per_tensor = 0.0, 0.0, {}
        for key, par, ref in zip(names, [p for layer in model.layers for p in (layer.weight, layer.bias)], params64):
            g64 = ref.grad
            d = par.grad.double() - g64
            tmax = float(d.abs().max()) / max(float(g64.abs().max()), 1e-300)
            trms = float(d.pow(2).mean().sqrt() / g64.pow(2).mean().sqrt().clamp_min(1e-300))
            per_tensor[key] = [tmax, trms]
            gmax, grms = max(gmax, tmax), max(grms, trms)
        row["worst_tensor_grad_max_abs_err_over_max_abs"] = gmax
        row["worst_tensor_grad_rms_err_over_rms"] = grms
        row["grad_err_per_tensor_max_rms"] = per_tensor
        rows[label(mode, products)] = row
    set_products(None)
    return {"model": "MLP(3, 4, num_layers=%d, num_channels=256)" % layers, "samples": n,
            "logit_scale": scale, "modes": rows}


def distance_table(dev, name, n=8192, seed=1):
    """Encoded models: each split mode against the exact-f32 kernels (same features bit for bit
    in bf16x6; hardware sin / cos in bf16x3)."""
    import fourier_feature_nets_amd as ffn
    torch.manual_seed(seed)
    if name == "tiny":
        model, views = ffn.PositionalFourierMLP(3, 4, 5.5).to(dev), None
    else:
        model = ffn.NeRF(8, 256, 9, 10, 3, 4, [4], True).to(dev)
        views = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=1)
    x = torch.rand(n, 3, device=dev) * 2 - 1
    probe = torch.randn(n, 4, device=dev) / n

    def run(mode, products):
        set_products(products)
        model.precision = model.train_precision = mode
        model.zero_grad()
        y = model(x) if views is None else model(x, views)
        (y * probe).sum().backward()
        return y.detach().double(), [p.grad.double().clone() for p in model.parameters() if p.requires_grad]

    base_y, base_g = run("f32", None)
    rows = {}
    for mode, products in MODES[1:]:
        y, g = run(mode, products)
        rows[label(mode, products)] = {
            "logits_max_abs_diff_over_max_abs": float((y - base_y).abs().max() / base_y.abs().max()),
            "worst_tensor_grad_max_abs_diff_over_max_abs": max(
                float((a - b).abs().max() / b.abs().max().clamp_min(1e-300)) for a, b in zip(g, base_g))}
    set_products(None)
    return {"model": name, "samples": n, "against": "the exact-f32 kernels", "modes": rows}


def timed(fn, iters=4):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def timing_table(dev, name, n):
    import fourier_feature_nets_amd as ffn
    from fourier_feature_nets_amd._lib import c_i64
    from fourier_feature_nets_amd.ops import _call, _dev
    torch.manual_seed(2)
    views = None
    if name == "mlp8":
        model = ffn.MLP(3, 4, num_layers=8, num_channels=256).to(dev)
    elif name == "tiny":
        model = ffn.PositionalFourierMLP(3, 4, 5.5).to(dev)
    else:
        model = ffn.NeRF(8, 256, 9, 10, 3, 4, [4], True).to(dev)
        views = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=1)
    prog = model.program()
    x = torch.rand(n, 3, device=dev) * 2 - 1
    saved = torch.empty((prog.saved_floats(n),), dtype=torch.float32, device=dev)
    d_logits = torch.randn(n, 4, device=dev) / n
    ws = prog.workspace(n)
    _, masks = prog._split_saved(saved, n)
    kblocks = sum((prog.fwd16.step[k].act_groups + prog.fwd16.step[k].aux_groups) // 2
                  for k in range(prog.fwd16.num_steps))
    flop = 2 * sum(spec.out * spec.ld for spec in prog.layers)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    rows = {}
    for mode, products in MODES:
        set_products(products)
        row = {}
        if mode == "bf16x3":
            row["inference_forward_ms"] = timed(lambda: prog.forward16(x, views))
        else:
            row["inference_forward_ms"] = timed(lambda: prog.forward(x, views, None, precision=mode))
        row["training_forward_ms"] = timed(lambda: prog.forward(x, views, saved, precision=mode))
        if mode == "f32":
            bwd = lambda: _call("ffn_mlp_backward_data", ctypes.byref(prog.bwd), _dev(prog.packed_bwd),
                                _dev(d_logits), c_i64(n), _dev(masks), _dev(ws.dz), c_i64(0), c_i64(0))
        elif mode == "bf16x3":
            bwd = lambda: _call("ffn_mlp_backward_data_bf16x3", ctypes.byref(prog.bwd16),
                                _dev(prog.packed16_bwd, torch.int16), _dev(d_logits), c_i64(n), _dev(masks),
                                _dev(ws.dz))
        else:
            bwd = lambda: _call("ffn_mlp_backward_data_bf16x6", ctypes.byref(prog.bwd_x6),
                                _dev(prog.packed_x6_bwd, torch.int16), _dev(d_logits), c_i64(n), _dev(masks),
                                _dev(ws.dz))
        row["backward_data_ms"] = timed(bwd)
        row = {k: round(v, 3) for k, v in row.items()}
        # per K block (16 K x 32 output channels x the pass's samples) and per CU, inference forward
        per_pass = {"f32": 128, "bf16x3": 128, "bf16x6": 64}[mode]      # samples a CU works on at a time
        passes_per_cu = n / per_pass / cus
        row["inference_ns_per_kblock_per_128_samples"] = round(
            row["inference_forward_ms"] * 1e6 / (passes_per_cu * kblocks) * (128 / per_pass), 1)
        row["inference_algorithmic_tflops"] = round(flop * n / row["inference_forward_ms"] / 1e9, 1)
        rows[label(mode, products)] = row
    set_products(None)
    base = rows["f32"]
    for key, row in rows.items():
        row["speedup_vs_f32"] = {k.replace("_ms", ""): round(base[k] / row[k], 3)
                                 for k in ("inference_forward_ms", "training_forward_ms", "backward_data_ms")}
    prog.release_workspaces()
    del saved, x, d_logits
    torch.cuda.empty_cache()
    return {"model": name, "samples": n, "k_blocks_per_pass_and_tile": kblocks,
            "algorithmic_flop_per_sample_forward": flop, "modes": rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--samples", type=int, default=1 << 22)
    ap.add_argument("--skip-timing", action="store_true")
    ap.add_argument("--error-seeds", type=int, default=0,
                    help="repeat the error table over this many seeds and report the spread of the ratios")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    doc = {"what": __doc__.split("\n\n")[0].replace("\n", " ")}
    try:
        doc["commit"] = open(os.path.join(ROOT, ".git_head")).read().split()[0]
    except OSError:
        doc["commit"] = None
    doc["errors_vs_float64"] = [error_table(dev, layers) for layers in (2, 4, 8)]
    if args.error_seeds:
        # seed-to-seed spread of the error ratios (split mode over exact kernels), per metric
        spread = {}
        for layers in (2, 8):
            ratios = {}
            for seed in range(1, args.error_seeds + 1):
                rows = error_table(dev, layers, n=4096, seed=seed)["modes"]
                for mode in ("bf16x6_6p", "bf16x6_9p"):
                    for key in ("logits_max_abs_err_over_max_abs", "logits_rms_err_over_rms",
                                "worst_tensor_grad_max_abs_err_over_max_abs", "worst_tensor_grad_rms_err_over_rms"):
                        ratios.setdefault("%s/%s" % (mode, key), []).append(rows[mode][key] / max(rows["f32"][key], 1e-300))
            spread["layers=%d" % layers] = {k: {"min": round(min(v), 3), "median": round(sorted(v)[len(v) // 2], 3),
                                                "max": round(max(v), 3)} for k, v in ratios.items()}
        doc["error_ratio_over_seeds"] = {"seeds": args.error_seeds, "ratios_split_over_exact": spread}
    doc["distance_from_exact_f32_kernels"] = [distance_table(dev, name) for name in ("tiny", "nerf")]
    if not args.skip_timing:
        doc["timings"] = [timing_table(dev, name, args.samples if name != "nerf" else args.samples // 2)
                          for name in ("mlp8", "tiny", "nerf")]
        t = {row["model"]: row for row in doc["timings"]}
        e = doc["errors_vs_float64"][-1]["modes"]
        speed = t["mlp8"]["modes"]["bf16x6_6p"]["speedup_vs_f32"]["inference_forward"]
        ratio = e["bf16x6_6p"]["logits_max_abs_err_over_max_abs"] / max(e["f32"]["logits_max_abs_err_over_max_abs"], 1e-300)
        doc["stop_rule"] = {"k_loop_speedup_vs_exact_f32 (mlp8 inference: eight 256-channel layers, no encoding)": speed,
                            "error_ratio_vs_exact_f32 (mlp8 logits, max abs)": round(ratio, 3),
                            "needs": ">= 1.25x at <= 2x", "verdict": "build" if speed >= 1.25 and ratio <= 2.0 else "stop"}
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
