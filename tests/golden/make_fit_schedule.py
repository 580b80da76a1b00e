"""The reference's `Raycaster.fit` ACROSS the removal of the centre crop (ray_caster.py:364-370:
at the first report with step >= crop_steps the three datasets go back to their mode, the step
counter advances and the epoch is abandoned for a fresh permutation) -> fit_schedule.npz.
training.npz's 12-step trajectory stays inside the crop phase, because the reference's
`_validate` needs > 102 400 rays per dataset outside it; this one runs on a 20 + 10 camera
128x128 rig (the PSNR scene of tests/psnr_parity.py) written in the reference's NPZ schema.
Build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fit_schedule.py

Recorded (data, not source): the initial weights, every training batch the reference drew (the
dataset indices handed to `_loss`), the training losses, the LogEntry table, the report lines."""

import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

TRAIN_CAMS, VAL_CAMS, SIZE, SAMPLES, BATCH = 20, 10, 128, 8, 256
NUM_STEPS, CROP_STEPS, REPORT = 14, 5, 5
MODEL = dict(num_channels=32, embedding_size=16)


def scene_file(path):
    from tests.psnr_ensemble import write_npz
    if not os.path.exists(path):
        write_npz(path, TRAIN_CAMS, VAL_CAMS, SIZE)
    return path


def main():
    sys.path.insert(0, HERE)
    from make_goldens import REFERENCE, _install_stubs
    _install_stubs()
    sys.path.insert(0, REFERENCE)
    sys.dont_write_bytecode = True
    import fourier_feature_nets as ffn
    npz = scene_file("/tmp/ffn_fit_schedule_scene.npz")
    torch.manual_seed(20080524)
    np.random.seed(20080524)
    torch.set_num_threads(4)
    model = ffn.PositionalFourierMLP(3, 4, 5.5, **MODEL)
    out = {"init/" + k: v.clone().numpy() for k, v in model.state_dict().items()}
    with contextlib.redirect_stdout(io.StringIO()):
        train = ffn.ImageDataset.load(npz, "train", SAMPLES, True, True, anneal_start=0.2, num_anneal_steps=8)
        val = ffn.ImageDataset.load(npz, "val", SAMPLES, True, False)
    torch.manual_seed(777)
    np.random.seed(777)
    caster = ffn.Raycaster(model)
    batches, losses, modes = [], [], []
    inner = caster._loss

    def spy(step, dataset, batch):
        value = inner(step, dataset, batch)
        if torch.is_grad_enabled() and value.requires_grad:
            batches.append(np.asarray(batch, np.int64))
            losses.append(float(value))
            modes.append(int(dataset.mode.value))
        return value

    caster._loss = spy
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        log = caster.fit(train, val, BATCH, 5e-4, NUM_STEPS, CROP_STEPS, REPORT, 0.1, 25000, 0.0, [],
                         disable_aml=True)
    print(buf.getvalue())
    out["batches"] = np.stack(batches)
    out["losses"] = np.array(losses, np.float64)
    out["modes"] = np.array(modes)
    out["stdout"] = np.array(buf.getvalue())
    out["log_steps"] = np.array([e.step for e in log])
    out["log_train_psnr"] = np.array([e.train_psnr for e in log])
    out["log_val_psnr"] = np.array([e.val_psnr for e in log])
    for key, value in model.state_dict().items():
        out["final/" + key] = value.numpy()
    np.savez_compressed(os.path.join(HERE, "fit_schedule.npz"), **out)
    print(len(batches), "training steps; modes", modes)


if __name__ == "__main__":
    main()
