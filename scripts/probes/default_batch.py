"""The reference drivers' DEFAULT batch (train_nerf.py:21-27: 1024 rays x 128 samples) through
TrainEngine.train_step the way fit drives it (epoch-level ray filter, no per-step sync):
ms/step and the per-sample rate relative to the large-batch step.  argv: model (tiny|nerf) steps"""
import contextlib, io, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench as B
import fourier_feature_nets_amd as ffn
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
R, S = 1024, 128
torch.manual_seed(0)
model = (ffn.NeRF(8, 256, 9, 10, 3, 4, [4], True) if name == "nerf" else ffn.PositionalFourierMLP(3, 4, 5.5)).to(dev)
intr, poses = B.synthetic_rig(20, 400)
cams = [ffn.CameraInfo.create("t%03d" % i, ffn.Resolution(400, 400), intr, p) for i, p in enumerate(poses)]
bounds = np.diag([2, 2, 2, 1]).astype(np.float32)
with contextlib.redirect_stdout(io.StringIO()):
    probe = ffn.RaySampler(bounds, cams, S, device=dev)
    images = B.analytic_images(probe)
    del probe
    ds = ffn.ImageDataset("train", images, bounds, cams, S, True, True, anneal_start=0.2, num_anneal_steps=2000, device=dev)
eng = ffn.TrainEngine(model, 0.0, None)
perm = torch.randperm(len(ds), device=dev)
def run(first, k, batch=R):
    rays, b = ds.epoch_ray_ids(perm[first * batch:(first + k) * batch], batch)
    for i in range(k):
        eng.train_step(ds, perm[(first + i) * batch:(first + i + 1) * batch], first + i, 5e-4, rays=rays[b[i]:b[i + 1]])
    return int(rays.numel())
run(0, 10); torch.cuda.synchronize()
t0 = time.perf_counter(); n_small = run(10, steps); torch.cuda.synchronize(); small = (time.perf_counter() - t0) / steps
big_batch = 32768
run(0, 2, big_batch); torch.cuda.synchronize()
t0 = time.perf_counter(); n_big = run(2, 6, big_batch); torch.cuda.synchronize(); big = (time.perf_counter() - t0) / 6
per_small, per_big = small * steps / n_small, big * 6 / n_big
print("%s default batch %d x %d: %.3f ms/step (%.0f valid rays/step), %.1f ns/ray; large batch %d: %.2f ms/step, %.1f ns/ray; ratio %.3f"
      % (name, R, S, small * 1e3, n_small / steps, per_small * 1e9, big_batch, big * 1e3, per_big * 1e9, per_big / per_small))
