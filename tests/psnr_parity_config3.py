"""PSNR parity for BASELINE config 3 -- the FULL NeRF (8 x 256, skip, view branch) trained with
opacity-guided focus sampling (64 stratified + 64 CDF-driven samples per ray from a frozen coarse
model; ray_sampler.py:234-357, :388-392; train_nerf.py's shape) -- by the protocol of
tests/psnr_parity.py that CAN pin 0.05 dB for a chaotic trajectory: re-synchronised segments.

    # oracle half: CPU (tens of minutes): trains K segments of N steps, writes the trajectory and
    # the weights + Adam moments at every checkpoint
    python -m tests.psnr_parity_config3 oracle --out profiles/r04_psnr_parity_config3_oracle.json \\
        --ckpt-dir tests/golden/_psnr_oracle_config3
    # HIP half: on the MI355X: (1) the oracle's weights at every checkpoint rendered by the HIP path
    # (same-weights PSNR), (2) from checkpoint k, N steps on the same ray batches / jitter / focus
    # draws -> checkpoint k+1's PSNR
    python -m tests.psnr_parity_config3 hip --oracle profiles/r04_psnr_parity_config3_oracle.json \\
        --ckpt-dir tests/golden/_psnr_oracle_config3 --out profiles/r04_psnr_parity_config3.json

Scene, batches, learning-rate schedule: tests/psnr_parity.py (100 + 7 cameras of 400x400, 1024
rays per step).  The coarse model is the tiny NeRF of that run after 300 steps
(tests/golden/_psnr_oracle/step_000300.pt), frozen; its CDF rows are computed per batch from its
current weights on both sides (the reference's table holds the same rows: the model does not
change).  Test infrastructure: drives the oracle, lives under tests/."""

import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tests.psnr_parity import BOUNDS, SEED, initial_model, lr_at, scene, step_rays, val_ids_of   # noqa: E402

S = 128
N_FOCUS = S - S // 2


def nerf_model():
    import fourier_feature_nets_amd as ffn
    torch.manual_seed(SEED + 1)
    return ffn.NeRF(8, 256, 9, 10, 3, 4, [4], True)     # train_nerf.py defaults


def coarse_weights(path):
    """(a, b, weights, biases) of the tiny coarse model from a tests/psnr_parity.py checkpoint."""
    model = initial_model()
    flat = torch.load(path)["params"]
    offset = 0
    for p in model._dense_params():
        n = p.numel()
        p.data.copy_(flat[offset:offset + n].view(p.shape))
        offset += n
    assert offset == flat.numel()
    return model


# ------------------------------------------------------------------------------- oracle half
def run_oracle(args):
    from oracle import ffn_oracle as orc
    torch.set_num_threads(args.threads or os.cpu_count())
    intr, poses, images, train_ids, val_ids = scene(args.cameras, args.val_cameras, args.size)
    st = orc.sampler_state(BOUNDS, [intr] * len(train_ids), [poses[i] for i in train_ids], args.size, args.size)
    sv = orc.sampler_state(BOUNDS, [intr] * len(val_ids), [poses[i] for i in val_ids], args.size, args.size)
    bad_t = np.zeros(st["num_rays"], bool)
    bad_t[st["invalid"]] = True
    bad_v = np.zeros(sv["num_rays"], bool)
    bad_v[sv["invalid"]] = True
    coarse = coarse_weights(args.coarse)
    coarse_ref = orc.OracleFourierMLP(coarse.a_values.data.clone(), coarse.b_values.data.clone(),
                                      [l.weight.data.clone() for l in coarse.layers],
                                      [l.bias.data.clone() for l in coarse.layers])
    model = nerf_model()
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    ref = orc.OracleNeRF(params, [4], True)
    trainer = orc.OracleTrainer(ref, 5e-4)
    names = [k for k, v in ref.p.items() if v.requires_grad]

    def gt(ids_list, ray_ids):
        img = images[ids_list]
        colors = torch.from_numpy(img[..., :3].astype(np.float32) / 255).reshape(-1, 3)
        alphas = torch.from_numpy(img[..., 3].astype(np.float32) / 255).reshape(-1)
        return orc.ground_truth(colors, alphas, torch.as_tensor(ray_ids))

    def sample(state, ids, noise, focus_u):
        """RaySampler.sample with focus sampling for rays `ids` (ray_sampler.py:359-403): the CDF
        rows are the coarse model's, from its probe pass over exactly these rays."""
        ids_t = torch.as_tensor(ids, dtype=torch.long)
        local = {"starts": state["starts"][ids_t], "directions": state["directions"][ids_t],
                 "near_far": state["near_far"][:, ids_t]}
        near, far = local["near_far"]
        t_probe = orc.linspace_rows(near, far, N_FOCUS)
        pos = local["starts"].unsqueeze(1) + t_probe.unsqueeze(2) * local["directions"].unsqueeze(1)
        with torch.no_grad():
            sigma = torch.nn.functional.softplus(coarse_ref(pos.reshape(-1, 3))[:, -1]).reshape(-1, N_FOCUS)
        cdf_rows = orc.determine_cdf(t_probe, sigma)
        return orc.sample(local, np.arange(len(ids)), None, S, noise=noise, cdfs=cdf_rows, focus_u=focus_u)

    vids = val_ids_of(sv["num_rays"], args.val_rays)
    vids = vids[~bad_v[vids]]
    unit_u = torch.linspace(0, 1, N_FOCUS).unsqueeze(0)
    vgc, vga = gt(val_ids, vids)

    def validate():
        with torch.no_grad():
            total, n = 0.0, 0
            for lo in range(0, len(vids), 1024):
                hi = min(lo + 1024, len(vids))
                pos, view, t, _ = sample(sv, vids[lo:hi], None, unit_u.repeat(hi - lo, 1))
                logits = ref(pos.reshape(-1, 3), view.reshape(-1, 3)).reshape(hi - lo, S, 4)
                color, alpha, _ = orc.render(logits, t, True)
                total += float(orc.mse_loss(color, alpha, vgc[lo:hi], vga[lo:hi], 0.1)) * (hi - lo)
                n += hi - lo
        return float(-10.0 * np.log10(total / n))

    os.makedirs(args.ckpt_dir, exist_ok=True)
    rows = []
    t0 = time.time()
    loss = None
    total_steps = args.segments * args.segment_steps
    for step in range(total_steps + 1):
        if step % args.segment_steps == 0:
            rows.append({"step": step, "val_psnr": validate(), "train_loss": loss})
            torch.save({"step": step, "count": trainer.count,
                        "params": {k: ref.p[k].detach().clone() for k in names},
                        "m": {k: m.clone() for k, m in zip(names, trainer.m)},
                        "v": {k: v.clone() for k, v in zip(names, trainer.v)}},
                       os.path.join(args.ckpt_dir, "step_%06d.pt" % step))
            print(rows[-1], "%.0f s" % (time.time() - t0), flush=True)
            with open(args.out, "w") as f:
                json.dump({"half": "oracle", "args": vars(args), "rows": rows, "complete": False}, f, indent=1)
        if step == total_steps:
            break
        ids = step_rays(step, st["num_rays"], args.rays)
        ids = ids[~bad_t[ids]]
        torch.manual_seed(args.noise_seed + step)
        noise = torch.rand((len(ids), S // 2))          # the sampler's draw order: jitter, then focus u
        focus_u = torch.rand((len(ids), N_FOCUS))
        pos, view, t, _ = sample(st, ids, noise, focus_u)
        gc, ga = gt(train_ids, ids)
        loss = trainer.step(pos, view, t, gc, ga, lr_at(step, decay_steps=250000))
        if step < 3 or step % 20 == 0:
            rows.append({"step": step, "batch": [int(len(ids)), int(ids.sum() % (1 << 31))], "loss": loss})
            print(rows[-1], "%.0f s" % (time.time() - t0), flush=True)
    torch.save({"params": torch.cat([p.detach().reshape(-1) for p in coarse._dense_params()])},
               os.path.join(args.ckpt_dir, "coarse.pt"))
    doc = {"half": "oracle", "args": vars(args), "rows": rows, "complete": True, "seconds": time.time() - t0,
           "threads": torch.get_num_threads(), "torch": torch.__version__, "valid_val_rays": int(len(vids)),
           "model": "NeRF(8, 256, 9, 10, 3, 4, [4], True)", "samples_per_ray": S,
           "coarse_model": "PositionalFourierMLP(3, 4, 5.5) after 300 steps of tests/psnr_parity.py, frozen"}
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    return doc


# ------------------------------------------------------------------------------- HIP half
def run_hip(args):
    import fourier_feature_nets_amd as ffn
    device = torch.device("cuda:0")
    with open(args.oracle) as f:
        ref = json.load(f)
    for key in ("cameras", "val_cameras", "size", "rays", "val_rays", "noise_seed", "segments", "segment_steps"):
        setattr(args, key, ref["args"][key])
    intr, poses, images, train_ids, val_ids = scene(args.cameras, args.val_cameras, args.size)
    coarse = initial_model()
    flat = torch.load(os.path.join(args.ckpt_dir, "coarse.pt"))["params"]
    offset = 0
    for p in coarse._dense_params():
        p.data.copy_(flat[offset:offset + p.numel()].view(p.shape))
        offset += p.numel()
    coarse = coarse.to(device)

    def dataset(ids, stratified, label):
        cams = [ffn.CameraInfo.create("%s%03d" % (label, i), ffn.Resolution(args.size, args.size), intr, poses[c])
                for i, c in enumerate(ids)]
        with contextlib.redirect_stdout(io.StringIO()):
            return ffn.ImageDataset(label, images[ids], BOUNDS, cams, S, True, stratified, coarse,
                                    device=device, focus_mode=args.focus_mode)

    train, val = dataset(train_ids, True, "train"), dataset(val_ids, False, "val")
    train.sampler.noise_source = "host"
    model = nerf_model().to(device)
    if args.precision != "f32":
        model.train_precision = model.precision = args.precision
    engine = ffn.TrainEngine(model)
    named = [(k, p) for k, p in model.named_parameters() if p.requires_grad]
    vids = torch.from_numpy(val_ids_of(val.sampler.num_rays, args.val_rays)).to(device)
    valid_val = int(val.ray_ids(vids).numel())

    def validate():
        model.eval()
        total, n = 0.0, 0
        with torch.no_grad():
            for lo in range(0, vids.numel(), 4096):
                chunk = vids[lo:lo + 4096]
                count = int(val.ray_ids(chunk).numel())
                total += float(engine.eval_loss(val, chunk, None)) * count
                n += count
        model.train()
        return float(-10.0 * np.log10(total / n))

    def load_state(path):
        blob = torch.load(path)
        # nn.Parameters are views of engine.flat; exp_avg / exp_avg_sq follow the same order
        offset = 0
        for key, p in named:
            n = p.numel()
            p.data.copy_(blob["params"][key].to(device))
            engine.exp_avg[offset:offset + n].copy_(blob["m"][key].reshape(-1).to(device))
            engine.exp_avg_sq[offset:offset + n].copy_(blob["v"][key].reshape(-1).to(device))
            offset += n
        assert offset == engine.flat.numel()
        engine.count = int(blob["count"])
        model.invalidate_packed()
        return int(blob["step"])

    theirs = {r["step"]: r["val_psnr"] for r in ref["rows"] if "val_psnr" in r}
    batches = {r["step"]: r["batch"] for r in ref["rows"] if "batch" in r}
    steps_sorted = sorted(theirs)
    same_weights, segments, mismatched = [], [], 0
    t0 = time.time()
    for a, b in zip(steps_sorted, steps_sorted[1:] + [None]):
        assert load_state(os.path.join(args.ckpt_dir, "step_%06d.pt" % a)) == a
        mine = validate()
        same_weights.append({"step": a, "hip": mine, "oracle": theirs[a], "delta_db": mine - theirs[a]})
        if b is None:
            break
        for step in range(a, b):
            ids = torch.from_numpy(step_rays(step, train.sampler.num_rays, args.rays)).to(device)
            if step in batches:
                got = train.ray_ids(ids)
                mismatched += int([int(got.numel()), int(got.sum().item() % (1 << 31))] != list(batches[step]))
            torch.manual_seed(args.noise_seed + step)
            engine.train_step(train, ids, None, lr_at(step, decay_steps=250000))
        mine = validate()
        segments.append({"from": a, "to": b, "hip": mine, "oracle": theirs[b], "delta_db": mine - theirs[b]})
        print(segments[-1], flush=True)
    engine.check_finite()
    git = None
    with contextlib.suppress(OSError):
        git = open(os.path.join(ROOT, ".git_head")).read().split()[0]
    doc = {"half": "hip", "precision": args.precision, "focus_mode": train.sampler.focus_mode,
           "args": vars(args), "commit": git,
           "model": ref.get("model"), "samples_per_ray": S, "coarse_model": ref.get("coarse_model"),
           "valid_val_rays": valid_val, "oracle_valid_val_rays": ref.get("valid_val_rays"),
           "same_weights": same_weights, "segments": segments,
           "same_weights_max_abs_delta_db": max(abs(r["delta_db"]) for r in same_weights),
           "segments_max_abs_delta_db": max(abs(r["delta_db"]) for r in segments),
           "batches_checked": len(batches), "batches_mismatched": mismatched, "bound_db": 0.05,
           "seconds": time.time() - t0, "device": torch.cuda.get_device_name(0)}
    doc["within_bound"] = bool(doc["same_weights_max_abs_delta_db"] < 0.05 and
                               doc["segments_max_abs_delta_db"] < 0.05 and mismatched == 0)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    return doc


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("half", choices=["oracle", "hip"])
    ap.add_argument("--segments", type=int, default=3)
    ap.add_argument("--segment-steps", type=int, default=100)
    ap.add_argument("--cameras", type=int, default=100)
    ap.add_argument("--val-cameras", type=int, default=7)
    ap.add_argument("--size", type=int, default=400)
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--val-rays", type=int, default=8192)
    ap.add_argument("--noise-seed", type=int, default=5000)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16x3"])
    ap.add_argument("--focus-mode", default="live", choices=["live", "table"])
    ap.add_argument("--coarse", default=os.path.join(ROOT, "tests", "golden", "_psnr_oracle", "step_000300.pt"))
    ap.add_argument("--oracle", help="(hip) the oracle half's trajectory")
    ap.add_argument("--ckpt-dir", required=True)
    ap.add_argument("--out", required=True)
    args = ap.parse_args(argv)
    doc = run_oracle(args) if args.half == "oracle" else run_hip(args)
    print(json.dumps({k: v for k, v in doc.items() if k not in ("rows", "args", "same_weights", "segments")}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
