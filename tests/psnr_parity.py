"""PSNR parity at the stated scale (BASELINE north_star: "rendered PSNR within 0.05 dB of the
reference"): the tiny NeRF on the benchmark's 100 x 400x400 synthetic scene, trained with the
reference's default batch (1024 rays x 64 samples, train_tiny_nerf.py) from the same weights, ray
batches and stratification noise by (a) the CPU oracle and (b) the HIP path; validation PSNR on
held-out cameras at fixed checkpoints.  Test infrastructure (it drives the oracle), hence under
tests/.  The two halves run in different places:

    # oracle half: CPU only, any machine (minutes to tens of minutes); writes the trajectory
    python -m tests.psnr_parity oracle --steps 1000 --out profiles/r03_psnr_parity_oracle.json
    # HIP half: on the MI355X; replays the same batches / noise and compares
    python -m tests.psnr_parity hip --oracle profiles/r03_psnr_parity_oracle.json \\
        --out profiles/r03_psnr_parity_400.json
    # HIP only, long run: final PSNR and rays/s
    python -m tests.psnr_parity hip --steps 5000 --out profiles/r03_psnr_5000_steps.json

Everything both halves must agree on is a pure function of the arguments: the camera rig and
the analytic RGBA images (float64 numpy from the camera matrices), the initial weights
(torch.manual_seed), the per-step ray ids (numpy RandomState) and the per-step jitter
(torch.manual_seed(noise_seed + step) followed by one torch.rand on the host -- the HIP sampler
draws it the same way with ``noise_source = "host"``)."""

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

SEED = 20080524            # the drivers' default seed (train_nerf.py:48)
BOUNDS = np.diag([2, 2, 2, 1]).astype(np.float32)


# ------------------------------------------------------------------------------- shared scene
def rig(num_cameras, size, fov_deg=40.0, distance=4.0, seed=SEED):
    """bench.py's synthetic rig: cameras on a seeded ring looking at the origin."""
    rng = np.random.RandomState(seed)
    focal = 0.5 * size / np.tan(0.5 * np.deg2rad(fov_deg))
    intr = np.array([[focal, 0, size / 2], [0, focal, size / 2], [0, 0, 1]], np.float32)
    poses = []
    for c in range(num_cameras):
        azi = 2 * np.pi * c / num_cameras
        alt = np.deg2rad(10 + 35 * rng.rand())
        eye = distance * np.array([np.cos(azi) * np.cos(alt), np.sin(alt), np.sin(azi) * np.cos(alt)])
        fwd = -eye / np.linalg.norm(eye)
        right = np.cross(fwd, np.array([0, 1.0, 0]))
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        pose = np.eye(4, dtype=np.float32)
        pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, down, fwd, eye
        poses.append(pose)
    return intr, poses


def analytic_image(intr, pose, size, radius=0.6):
    """RGBA uint8 image of a normal-shaded sphere at the origin, in float64 from the camera
    matrices alone (so that both halves hold byte-identical ground truth)."""
    k = intr.astype(np.float64)
    e = pose.astype(np.float64)
    xs, ys = np.meshgrid(np.arange(size, dtype=np.float64), np.arange(size, dtype=np.float64))
    cam = np.stack([(xs - k[0, 2]) / k[0, 0], (ys - k[1, 2]) / k[1, 1], np.ones_like(xs)], -1)
    d = cam @ e[:3, :3].T
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    o = e[:3, 3]
    b = d @ o
    c = o @ o - radius * radius
    disc = b * b - c
    hit = disc > 0
    t = -b - np.sqrt(np.clip(disc, 0, None))
    normal = o + t[..., None] * d
    normal /= np.maximum(np.linalg.norm(normal, axis=-1, keepdims=True), 1e-30)
    rgb = (0.5 + 0.5 * normal) * hit[..., None]
    rgba = np.concatenate([rgb, hit[..., None].astype(np.float64)], -1)
    return (rgba * 255).astype(np.uint8)


def scene(train_cams, val_cams, size):
    intr, poses = rig(train_cams + val_cams, size)
    # held-out cameras: every (train+val)/val-th camera of the ring
    stride = (train_cams + val_cams) / val_cams
    val_ids = sorted({int(i * stride + stride / 2) for i in range(val_cams)})
    train_ids = [i for i in range(train_cams + val_cams) if i not in val_ids]
    images = np.stack([analytic_image(intr, p, size) for p in poses])
    return intr, poses, images, train_ids, val_ids


def initial_model():
    import fourier_feature_nets_amd as ffn
    torch.manual_seed(SEED)
    return ffn.PositionalFourierMLP(3, 4, 5.5)          # train_tiny_nerf.py "positional" defaults


def step_rays(step, num_rays, count, seed=SEED):
    return np.random.RandomState(seed + 7919 * (step + 1)).randint(0, num_rays, count).astype(np.int64)


def lr_at(step, lr0=5e-4, rate=0.1, decay_steps=25000):
    return lr0 * rate ** (step / decay_steps)


def val_ids_of(num_val_rays, count):
    return np.linspace(0, num_val_rays, count, endpoint=False).astype(np.int64)


# ------------------------------------------------------------------------------- oracle half
def run_oracle(args):
    from oracle import ffn_oracle as orc
    torch.set_num_threads(args.threads or os.cpu_count())
    intr, poses, images, train_ids, val_ids = scene(args.cameras, args.val_cameras, args.size)
    t0 = time.time()
    st = orc.sampler_state(BOUNDS, [intr] * len(train_ids), [poses[i] for i in train_ids], args.size, args.size)
    sv = orc.sampler_state(BOUNDS, [intr] * len(val_ids), [poses[i] for i in val_ids], args.size, args.size)
    print("oracle ray state: %.1f s" % (time.time() - t0), flush=True)
    bad_t = np.zeros(st["num_rays"], bool)
    bad_t[st["invalid"]] = True
    bad_v = np.zeros(sv["num_rays"], bool)
    bad_v[sv["invalid"]] = True

    def gt(ids_list, ray_ids):
        img = images[ids_list]
        colors = torch.from_numpy(img[..., :3].astype(np.float32) / 255).reshape(-1, 3)
        alphas = torch.from_numpy(img[..., 3].astype(np.float32) / 255).reshape(-1)
        return orc.ground_truth(colors, alphas, torch.as_tensor(ray_ids))

    model = initial_model()
    ref = orc.OracleFourierMLP(model.a_values.data.clone(), model.b_values.data.clone(),
                               [l.weight.data.clone() for l in model.layers],
                               [l.bias.data.clone() for l in model.layers])
    trainer = orc.OracleTrainer(ref, 5e-4)
    vids = val_ids_of(sv["num_rays"], args.val_rays)
    vids = vids[~bad_v[vids]]
    vpos, _, vt, _ = orc.sample(sv, vids, None, args.samples)
    vgc, vga = gt(val_ids, vids)

    def validate():
        with torch.no_grad():
            total, n = 0.0, 0
            for lo in range(0, len(vids), 4096):
                hi = min(lo + 4096, len(vids))
                logits = ref(vpos[lo:hi].reshape(-1, 3)).reshape(hi - lo, args.samples, 4)
                color, alpha, _ = orc.render(logits, vt[lo:hi], True)
                total += float(orc.mse_loss(color, alpha, vgc[lo:hi], vga[lo:hi], 0.1)) * (hi - lo)
                n += hi - lo
        return float(-10.0 * np.log10(total / n))

    rows = []
    t0 = time.time()
    loss = None
    if args.ckpt_dir:
        os.makedirs(args.ckpt_dir, exist_ok=True)
    for step in range(args.steps + 1):
        if step % args.every == 0 or step == args.steps:
            rows.append({"step": step, "val_psnr": validate(), "train_loss": loss})
            if args.ckpt_dir:       # weights + Adam moments: the state a re-synchronised run starts from
                flat = lambda ts: torch.cat([t.detach().reshape(-1) for t in ts])      # noqa: E731
                torch.save({"step": step, "count": trainer.count, "params": flat(ref.parameters()),
                            "m": flat(trainer.m), "v": flat(trainer.v)},
                           os.path.join(args.ckpt_dir, "step_%06d.pt" % step))
            print(rows[-1], "%.0f s" % (time.time() - t0), flush=True)
            with open(args.out, "w") as f:
                json.dump({"half": "oracle", "args": vars(args), "rows": rows, "complete": False}, f, indent=1)
        if step == args.steps:
            break
        ids = step_rays(step, st["num_rays"], args.rays)
        ids = ids[~bad_t[ids]]
        torch.manual_seed(args.noise_seed + step)
        noise = torch.rand((len(ids), args.samples))
        pos, view, t, _ = orc.sample(st, ids, None, args.samples, noise=noise)
        gc, ga = gt(train_ids, ids)
        loss = trainer.step(pos, view, t, gc, ga, lr_at(step))
        rows_meta = (int(len(ids)), int(ids.sum() % (1 << 31)))
        if step < 4 or step % args.every == 0:
            rows.append({"step": step, "batch": rows_meta, "loss": loss})
    checksum = float(sum(float(w.detach().double().abs().sum()) for w in ref.weights))
    doc = {"half": "oracle", "args": vars(args), "rows": rows, "complete": True,
           "weights_abs_sum": checksum, "seconds": time.time() - t0,
           "threads": torch.get_num_threads(), "torch": torch.__version__, "numpy": np.__version__,
           "valid_val_rays": int(len(vids))}
    if args.oracle:             # control: drift between two oracle runs
        with open(args.oracle) as f:
            other = json.load(f)
        theirs = {r["step"]: r["val_psnr"] for r in other["rows"] if "val_psnr" in r}
        diffs = [{"step": r["step"], "this": r["val_psnr"], "other": theirs[r["step"]],
                  "delta_db": r["val_psnr"] - theirs[r["step"]]}
                 for r in rows if "val_psnr" in r and r["step"] in theirs]
        doc["control_against"] = {"file": args.oracle, "threads": other.get("threads"), "comparison": diffs,
                                  "max_abs_delta_db": max(abs(d["delta_db"]) for d in diffs)}
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    return doc


# ------------------------------------------------------------------------------- HIP half
def run_hip(args):
    import fourier_feature_nets_amd as ffn
    device = torch.device("cuda:0")
    ref = None
    if args.oracle:
        with open(args.oracle) as f:
            ref = json.load(f)
        for key in ("cameras", "val_cameras", "size", "samples", "rays", "val_rays", "every", "noise_seed"):
            setattr(args, key, ref["args"][key])
        if args.steps is None:
            args.steps = max(r["step"] for r in ref["rows"] if "val_psnr" in r)
    intr, poses, images, train_ids, val_ids = scene(args.cameras, args.val_cameras, args.size)

    def dataset(ids, stratified, label):
        cams = [ffn.CameraInfo.create("%s%03d" % (label, i), ffn.Resolution(args.size, args.size), intr, poses[c])
                for i, c in enumerate(ids)]
        with contextlib.redirect_stdout(io.StringIO()):
            return ffn.ImageDataset(label, images[ids], BOUNDS, cams, args.samples, True, stratified,
                                    device=device)

    train, val = dataset(train_ids, True, "train"), dataset(val_ids, False, "val")
    train.sampler.noise_source = "host"
    model = initial_model().to(device)
    if args.precision != "f32":
        model.train_precision = model.precision = args.precision
    engine = ffn.TrainEngine(model)
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
        engine.flat.copy_(blob["params"].to(device))
        engine.exp_avg.copy_(blob["m"].to(device))
        engine.exp_avg_sq.copy_(blob["v"].to(device))
        engine.count = int(blob["count"])
        model.invalidate_packed()
        return int(blob["step"])

    def train_range(first, last):
        for step in range(first, last):
            ids = torch.from_numpy(step_rays(step, train.sampler.num_rays, args.rays)).to(device)
            torch.manual_seed(args.noise_seed + step)
            engine.train_step(train, ids, None, lr_at(step))

    resync = None
    if args.ckpt_dir and ref is not None:
        # (1) the SAME model rendered by both paths: the oracle's weights at every checkpoint,
        # validation PSNR of the HIP path against the oracle's own number
        # (2) re-synchronised segments: from the oracle's weights + Adam moments at checkpoint k the
        # HIP path trains to checkpoint k+1 on the same batches and noise
        theirs = {r["step"]: r["val_psnr"] for r in ref["rows"] if "val_psnr" in r}
        steps_sorted = sorted(theirs)
        same_weights, segments = [], []
        for a, b in zip(steps_sorted, steps_sorted[1:] + [None]):
            path = os.path.join(args.ckpt_dir, "step_%06d.pt" % a)
            if not os.path.exists(path):
                continue
            assert load_state(path) == a
            mine = validate()
            same_weights.append({"step": a, "hip": mine, "oracle": theirs[a], "delta_db": mine - theirs[a]})
            if b is not None:
                train_range(a, b)
                mine = validate()
                segments.append({"from": a, "to": b, "hip": mine, "oracle": theirs[b], "delta_db": mine - theirs[b]})
        resync = {"same_weights": same_weights, "segments": segments,
                  "same_weights_max_abs_delta_db": max(abs(r["delta_db"]) for r in same_weights),
                  "segments_max_abs_delta_db": max(abs(r["delta_db"]) for r in segments)}
        load_state(os.path.join(args.ckpt_dir, "step_%06d.pt" % steps_sorted[0]))

    rows, mismatched = [], 0
    expect = {r["step"]: r for r in (ref["rows"] if ref else []) if "batch" in r}
    torch.cuda.synchronize()
    t0 = time.time()
    train_seconds = 0.0
    loss = None
    for step in range(args.steps + 1):
        if step % args.every == 0 or step == args.steps:
            torch.cuda.synchronize()
            tv = time.time()
            rows.append({"step": step, "val_psnr": validate(), "train_loss": loss})
            train_seconds -= time.time() - tv
        if step == args.steps:
            break
        ids = torch.from_numpy(step_rays(step, train.sampler.num_rays, args.rays)).to(device)
        if step in expect:
            mine = train.ray_ids(ids)
            meta = [int(mine.numel()), int(mine.sum().item() % (1 << 31))]
            mismatched += int(meta != list(expect[step]["batch"]))
        torch.manual_seed(args.noise_seed + step)
        out = engine.train_step(train, ids, None, lr_at(step))
        if step % args.every == 0 or step == args.steps - 1:
            loss = float(out)
    torch.cuda.synchronize()
    train_seconds += time.time() - t0
    engine.check_finite()
    doc = {"half": "hip", "precision": args.precision, "args": vars(args), "rows": rows,
           "valid_val_rays": valid_val, "train_seconds_excl_validation": train_seconds,
           "rays_per_s_incl_host_overheads": args.rays * args.steps / max(train_seconds, 1e-9),
           "device": torch.cuda.get_device_name(0)}
    if ref is not None:
        theirs = {r["step"]: r["val_psnr"] for r in ref["rows"] if "val_psnr" in r}
        diffs = [{"step": r["step"], "hip": r["val_psnr"], "oracle": theirs[r["step"]],
                  "delta_db": r["val_psnr"] - theirs[r["step"]]} for r in rows if r["step"] in theirs]
        doc["comparison"] = diffs
        doc["max_abs_delta_db"] = max(abs(d["delta_db"]) for d in diffs)
        doc["final_delta_db"] = diffs[-1]["delta_db"]
        doc["batches_checked"] = len(expect)
        doc["batches_mismatched"] = mismatched
        doc["oracle_valid_val_rays"] = ref.get("valid_val_rays")
        doc["bound_db"] = 0.05
        doc["free_running_within_bound"] = bool(doc["max_abs_delta_db"] < 0.05)
        if resync is not None:
            doc["resynchronised"] = resync
            doc["within_bound"] = bool(resync["same_weights_max_abs_delta_db"] < 0.05 and
                                       resync["segments_max_abs_delta_db"] < 0.05)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(doc, f, indent=1)
    return doc


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("half", choices=["oracle", "hip"])
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--every", type=int, default=100, help="validation interval")
    ap.add_argument("--cameras", type=int, default=100)
    ap.add_argument("--val-cameras", type=int, default=7)
    ap.add_argument("--size", type=int, default=400)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--val-rays", type=int, default=16384)
    ap.add_argument("--noise-seed", type=int, default=1000)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16x3"])
    ap.add_argument("--oracle", help="(hip) trajectory written by the oracle half (or, for the oracle half: a "
                                     "previous oracle trajectory to report the divergence from -- the control "
                                     "for how far two runs of the SAME arithmetic drift apart when only the "
                                     "summation order changes, e.g. --threads 4 against --threads 8)")
    ap.add_argument("--ckpt-dir", help="oracle: write weights + Adam moments at every checkpoint; hip: "
                                       "replay re-synchronised segments from them")
    ap.add_argument("--out", required=True)
    args = ap.parse_args(argv)
    if args.half == "oracle":
        if args.steps is None:
            args.steps = 1000
        doc = run_oracle(args)
    else:
        if args.steps is None and not args.oracle:
            args.steps = 5000
        doc = run_hip(args)
    print(json.dumps({k: v for k, v in doc.items() if k not in ("rows", "comparison", "args")}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
