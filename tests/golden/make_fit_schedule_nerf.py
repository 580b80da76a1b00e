"""BASELINE config 3 as a whole, pinned by the reference's OWN `fit`: a full NeRF (view branch,
skip connection) trained with opacity-guided sampling (train_nerf.py:85-141: `ImageDataset.load(...,
opacity_model, ...)` for both datasets, ray_sampler.py:148-166 builds one CDF row per ray from a
coarse pass at construction, ray_sampler.py:301-357,388-392 draws half of every ray's samples from
it) through `Raycaster.fit` across the crop removal -> fit_schedule_nerf.npz.  Same 20 + 10 camera
128x128 rig as make_fit_schedule.py.  Build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fit_schedule_nerf.py

Recorded (data, not source): the opacity model's volume, the NeRF's initial weights, checksums and
a strided slice of the two samplers' CDF tables (valid rays), every training batch the reference drew, the
t-values the sampler handed to every training step, the training losses, the LogEntry table, the
report lines, the final weights."""

import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

TRAIN_CAMS, VAL_CAMS, SIZE, SAMPLES, BATCH = 20, 10, 128, 16, 256
NUM_STEPS, CROP_STEPS, REPORT = 14, 5, 5
ANNEAL_START, ANNEAL_STEPS = 0.2, 8
NERF = dict(num_layers=4, num_channels=64, max_log_scale_pos=9.0, num_freq_pos=10,
            max_log_scale_view=3.0, num_freq_view=4, skips=[2], include_inputs=True)
VOXEL_SIDE, VOXEL_SCALE = 12, 1.5
CDF_STRIDE = 997


def voxel_volume(seed=4242):
    """A seeded opacity volume: a soft ball of density (logit +3 inside radius 0.9 of the cube
    [-1.5, 1.5]^3, falling to -6 outside) plus seeded noise, so the CDF rows are neither flat nor
    one-hot.  Returns the (1,4,S,S,S) float32 array put into `Voxels.voxels`."""
    rng = np.random.RandomState(seed)
    axis = (np.arange(VOXEL_SIDE, dtype=np.float32) + 0.5) / VOXEL_SIDE * 2 - 1
    z, y, x = np.meshgrid(axis, axis, axis, indexing="ij")
    radius = np.sqrt(x * x + y * y + z * z) * VOXEL_SCALE
    sigma = np.clip((0.9 - radius) * 12.0, -4.0, 5.0) + rng.randn(*radius.shape).astype(np.float32) * 0.7
    volume = rng.randn(1, 4, VOXEL_SIDE, VOXEL_SIDE, VOXEL_SIDE).astype(np.float32) * 0.3
    volume[0, 3] = sigma
    return volume.astype(np.float32)


def scene_file(path):
    from tests.psnr_ensemble import write_npz
    if not os.path.exists(path):
        write_npz(path, TRAIN_CAMS, VAL_CAMS, SIZE)
    return path


def cdf_summary(sampler):
    """What the fixture keeps of a sampler's (rays, S_f - 1) CDF table, over the rays that hit the
    volume (rows of `invalid_rays` are never sampled): the float64 sum, the float64 sum of
    squares, and every CDF_STRIDE-th valid row with its ray id."""
    cdfs = sampler.cdfs
    valid = np.ones(len(cdfs), bool)
    valid[sorted(sampler.invalid_rays)] = False
    ids = np.nonzero(valid)[0]
    c = cdfs[torch.from_numpy(ids)].double()
    pick = ids[::CDF_STRIDE]
    return (np.array([float(c.sum()), float((c * c).sum()), float(len(ids))]), pick.astype(np.int64),
            cdfs[torch.from_numpy(pick)].numpy().copy())


def main():
    sys.path.insert(0, HERE)
    from make_goldens import REFERENCE, _install_stubs
    _install_stubs()
    sys.path.insert(0, REFERENCE)
    sys.dont_write_bytecode = True
    import fourier_feature_nets as ffn
    assert os.path.realpath(ffn.__file__).startswith(REFERENCE), ffn.__file__
    npz = scene_file("/tmp/ffn_fit_schedule_scene.npz")
    torch.manual_seed(20080524)
    np.random.seed(20080524)
    torch.set_num_threads(4)
    model = ffn.NeRF(**NERF)
    out = {"init/" + k: v.clone().numpy() for k, v in model.state_dict().items()}
    opacity = ffn.Voxels(VOXEL_SIDE, VOXEL_SCALE)
    with torch.no_grad():
        opacity.voxels.copy_(torch.from_numpy(voxel_volume()))
    out["opacity/voxels"] = opacity.voxels.detach().numpy().copy()
    out["opacity/bias"] = opacity.bias.detach().numpy().copy()
    with contextlib.redirect_stdout(io.StringIO()):
        train = ffn.ImageDataset.load(npz, "train", SAMPLES, True, True, opacity, BATCH, "RGB",
                                      anneal_start=ANNEAL_START, num_anneal_steps=ANNEAL_STEPS)
        val = ffn.ImageDataset.load(npz, "val", SAMPLES, True, False, opacity, BATCH, "RGB")
    out["train_cdf_sums"], out["train_cdf_ids"], out["train_cdf_rows"] = cdf_summary(train.sampler)
    out["val_cdf_sums"], out["val_cdf_ids"], out["val_cdf_rows"] = cdf_summary(val.sampler)
    torch.manual_seed(777)
    np.random.seed(777)
    caster = ffn.Raycaster(model)
    batches, losses, modes, t_values = [], [], [], []
    inner = caster._loss
    sampler_sample = type(train.sampler).sample
    last_t = {}

    def spy_sample(self, idx, step):
        samples = sampler_sample(self, idx, step)
        last_t["t"] = samples.t_values.clone().numpy()
        return samples

    type(train.sampler).sample = spy_sample

    def spy(step, dataset, batch):
        value = inner(step, dataset, batch)
        if torch.is_grad_enabled() and value.requires_grad:
            batches.append(np.asarray(batch, np.int64))
            losses.append(float(value))
            modes.append(int(dataset.mode.value))
            t_values.append(last_t["t"])
        return value

    caster._loss = spy
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            log = caster.fit(train, val, BATCH, 5e-4, NUM_STEPS, CROP_STEPS, REPORT, 0.1, 25000, 0.0, [],
                             disable_aml=True)
    finally:
        type(train.sampler).sample = sampler_sample
    print(buf.getvalue())
    out["batches"] = np.stack(batches)
    out["losses"] = np.array(losses, np.float64)
    out["modes"] = np.array(modes)
    out["t_rows"] = np.array([len(t) for t in t_values])
    out["t_values"] = np.concatenate(t_values)
    out["stdout"] = np.array(buf.getvalue())
    out["log_steps"] = np.array([e.step for e in log])
    out["log_train_psnr"] = np.array([e.train_psnr for e in log])
    out["log_val_psnr"] = np.array([e.val_psnr for e in log])
    for key, value in model.state_dict().items():
        out["final/" + key] = value.numpy()
    np.savez_compressed(os.path.join(HERE, "fit_schedule_nerf.npz"), **out)
    print(len(batches), "training steps; modes", modes, "t rows", out["t_rows"].tolist())


if __name__ == "__main__":
    main()
