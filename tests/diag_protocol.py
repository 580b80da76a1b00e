"""Diagnostic for a paired-ensemble protocol: the first optimisation step of seed 0 in detail
(validation PSNR of the INITIAL weights on both validation sets, the first training batch, its
loss, validation after the step), for the reference (CPU, build container) or the HIP path.
    python -m tests.diag_protocol reference|hip --out file.json [protocol flags of psnr_ensemble]"""
import argparse, contextlib, io, json, os, sys
import numpy as np, torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import psnr_ensemble as pe          # noqa: E402
from tests.psnr_parity import SEED              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("half", choices=["reference", "hip"])
    ap.add_argument("--out", required=True)
    ap.add_argument("--fit", type=int, default=0)
    args = ap.parse_args()
    ns = argparse.Namespace(model="nerf", opacity="voxels", voxel_side=32, size=128, cameras=20, val_cameras=4,
                            samples=128, rays=1024, crop_steps=1000, anneal_steps=150, workdir="/tmp/ffn_diag")
    os.makedirs(ns.workdir, exist_ok=True)
    npz = os.path.join(ns.workdir, "scene.npz")
    if not os.path.exists(npz):
        pe.write_npz(npz, ns.cameras, ns.val_cameras, ns.size)
    if args.half == "reference":
        from tests.golden.make_goldens import REFERENCE, _install_stubs
        _install_stubs()
        sys.path.insert(0, REFERENCE)
        import fourier_feature_nets as ffn
        device, kwargs = None, {}
        torch.set_num_threads(4)
    else:
        import fourier_feature_nets_amd as ffn
        device = torch.device("cuda", 0)
        kwargs = {"device": device, "focus_mode": "table"}
    torch.manual_seed(SEED)
    np.random.seed(SEED % (2 ** 32))
    model = ffn.NeRF(8, 256, 9.0, 10, 3.0, 4, [4], True)
    opacity = ffn.Voxels(32, 1.0)
    with torch.no_grad():
        opacity.voxels.copy_(torch.from_numpy(pe.ball_volume(32)))
    if device is not None:
        opacity, model = opacity.to(device), model.to(device)
    with contextlib.redirect_stdout(io.StringIO()):
        train = ffn.ImageDataset.load(npz, "train", ns.samples, True, True, opacity, 4096, "RGB",
                                      anneal_start=0.2, num_anneal_steps=ns.anneal_steps, **kwargs)
        val = ffn.ImageDataset.load(npz, "val", ns.samples, True, False, opacity, 4096, "RGB", **kwargs)
    if device is not None:
        train.sampler.noise_source = "host"
    out, arrays = {}, {}
    caster = ffn.Raycaster(model)
    state = torch.get_rng_state()
    out["rng_probe"] = float(torch.rand(1))
    torch.set_rng_state(state)
    if args.fit:
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            caster.fit(train, val, ns.rays, 5e-4, args.fit, ns.crop_steps, 100, 0.1, 25000, 0.0, [], True)
        out["fit_lines"] = [l for l in buf.getvalue().splitlines() if "psnr" in l]
        out["rng_probe_after"] = float(torch.rand(1))
        out["weight_sums"] = {k: float(v.double().abs().sum()) for k, v in model.state_dict().items()}
        print(json.dumps(out, indent=1))
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)
        return
    with contextlib.redirect_stdout(io.StringIO()):
        trainval = train.sample_cameras(val.num_cameras, val.num_samples, False)
    out["trainval_cameras"] = [c.name for c in trainval.cameras] if hasattr(trainval, "cameras") else None
    Mode = ffn.RayDataset.Mode
    for ds in (train, val, trainval):
        ds.mode = Mode.Center
    out["len"] = [len(train), len(val), len(trainval)]
    cd = train.sampler.cdfs
    out["train_cdf_sum"] = float(cd.double().sum())
    out["val_cdf_sum"] = float(val.sampler.cdfs.double().sum())
    out["trainval_cdf_sum"] = float(trainval.sampler.cdfs.double().sum())
    if args.half == "reference":
        out["val_psnr_init"] = float(caster._validate(val, ns.rays, 0))
        out["trainval_psnr_init"] = float(caster._validate(trainval, ns.rays, 0))
        # one batch through _loss by hand
        index = np.arange(len(train))
        np.random.shuffle(index)
        batch = index[:ns.rays].tolist()
        out["first_batch_head"] = batch[:8]
        torch.manual_seed(123)
        rays = train.get_rays(batch, 0)
        arrays["batch"] = np.asarray(batch)
        out["first_t_sum"] = float(rays.t_values.double().sum())
        out["first_rows"] = int(rays.t_values.shape[0])
    else:
        engine = ffn.TrainEngine(model)
        out["val_psnr_init"] = float(caster._validate(engine, val, ns.rays, 0))
        out["trainval_psnr_init"] = float(caster._validate(engine, trainval, ns.rays, 0))
        order = caster._epoch_order(len(train), engine)
        batch = order[:ns.rays]
        out["first_batch_head"] = batch[:8].cpu().tolist()
        torch.manual_seed(123)
        rays = train.get_rays(batch, 0)
        out["first_t_sum"] = float(rays.t_values.double().sum())
        out["first_rows"] = int(rays.t_values.shape[0])
        arrays["batch"] = batch.cpu().numpy()
    arrays["t"] = rays.t_values.detach().cpu().numpy()
    arrays["train_cdf_rowsum"] = train.sampler.cdfs.double().sum(-1).cpu().numpy()
    arrays["val_cdf_rowsum"] = val.sampler.cdfs.double().sum(-1).cpu().numpy()
    arrays["train_invalid"] = np.array(sorted(int(i) for i in train.sampler.invalid_rays)) \
        if not torch.is_tensor(train.sampler.invalid_rays) else train.sampler.invalid_rays.cpu().numpy()
    arrays["near_far"] = train.sampler.near_far.cpu().numpy()
    np.savez_compressed(args.out.replace(".json", ".npz"), **arrays)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
