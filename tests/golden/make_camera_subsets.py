"""Camera subsets picked by the REFERENCE's own ``RayDataset.sample_cameras`` (ray_dataset.py:185-216)
for a handful of rigs -> camera_subsets.json.  Build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_camera_subsets.py

The method is run unbound on a stand-in object that carries what it reads (``num_cameras``,
``sampler.cameras[i].position``, ``label``) and whose ``subset`` returns the camera list it was
given -- the list ORDER is the fixture (the reference iterates a Python set).  Only rig parameters
and index lists are recorded (data, not source)."""

import json
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def rig_positions(kind, count, seed):
    """(count, 3) float32 camera positions: the PSNR scene's rig, a ring (many ties) or noise."""
    if kind == "psnr_scene":
        from tests.psnr_parity import scene
        _, poses, _, train_ids, _ = scene(count, 7, 8)
        return np.stack([np.asarray(poses[i], np.float32)[:3, 3] for i in train_ids])
    if kind == "ring":
        a = np.arange(count) * (2 * np.pi / count)
        return np.stack([4 * np.cos(a), np.zeros(count), 4 * np.sin(a)], 1).astype(np.float32)
    rng = np.random.default_rng(seed)
    return rng.normal(size=(count, 3)).astype(np.float32)


CASES = [("psnr_scene", 100, 0, 7), ("psnr_scene", 20, 0, 5), ("ring", 64, 0, 8), ("ring", 100, 0, 7),
         ("ring", 12, 0, 12), ("noise", 40, 1, 6), ("noise", 200, 2, 10), ("noise", 300, 3, 25),
         ("noise", 5, 4, 7)]


def main():
    sys.path.insert(0, HERE)
    from make_goldens import REFERENCE, _install_stubs
    _install_stubs()
    sys.path.insert(0, REFERENCE)
    sys.dont_write_bytecode = True
    import fourier_feature_nets as ref
    out = []
    for kind, count, seed, pick in CASES:
        pos = rig_positions(kind, count, seed)
        cams = [SimpleNamespace(position=p[None, :]) for p in pos]
        stand_in = SimpleNamespace(num_cameras=count, sampler=SimpleNamespace(cameras=cams), label="x",
                                   subset=lambda cameras, *_: [int(c) for c in cameras])
        chosen = ref.RayDataset.sample_cameras(stand_in, pick, 64, False)
        out.append({"rig": kind, "cameras": count, "seed": seed, "pick": pick, "chosen": chosen})
        print(kind, count, pick, chosen)
    with open(os.path.join(HERE, "camera_subsets.json"), "w") as f:
        json.dump({"source": "matajoh/fourier_feature_nets v1.0.0 RayDataset.sample_cameras", "cases": out}, f, indent=1)


if __name__ == "__main__":
    main()
