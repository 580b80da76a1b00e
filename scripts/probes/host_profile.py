"""Where the HOST spends a default-batch step (1024 rays x 128 samples, tiny NeRF): cProfile over
300 TrainEngine.train_step calls driven like `fit` (epoch-level validity filter).
   python scripts/probes/host_profile.py [tiny|nerf]"""
import contextlib, cProfile, io, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.argv = sys.argv[:2]
import bench
import fourier_feature_nets_amd as ffn
name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
device = torch.device("cuda", 0)
intr, poses = bench.synthetic_rig(20, 400)
cams = [ffn.CameraInfo.create("train%03d" % i, ffn.Resolution(400, 400), intr, p) for i, p in enumerate(poses)]
bounds = np.diag([2, 2, 2, 1]).astype(np.float32)
with contextlib.redirect_stdout(io.StringIO()):
    probe = ffn.RaySampler(bounds, cams, 128, device=device)
    images = bench.analytic_images(probe)
    del probe
    ds = ffn.ImageDataset("train", images, bounds, cams, 128, True, True, anneal_start=0.2, num_anneal_steps=2000, device=device)
torch.manual_seed(1)
model = (ffn.PositionalFourierMLP(3, 4, 5.5) if name == "tiny" else ffn.NeRF(8, 256, 9, 10, 3, 4, [4], True)).to(device)
engine = ffn.TrainEngine(model, 0.0, None)
count, batch = 300, 1024
index = torch.randint(0, len(ds), (count * batch,), device=device)
ids, cuts = ds.epoch_ray_ids(index, batch)

def run():
    for i in range(count):
        engine.train_step(ds, index[i * batch:(i + 1) * batch], 1000 + i, 5e-4, rays=ids[cuts[i]:cuts[i + 1]])

run()
torch.cuda.synchronize()
t0 = time.perf_counter()
run()
enq = time.perf_counter() - t0
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("%s: %.4f ms/step, host enqueue %.4f ms/step" % (name, 1e3 * tot / count, 1e3 * enq / count))
prof = cProfile.Profile()
prof.enable()
run()
prof.disable()
torch.cuda.synchronize()
out = io.StringIO()
pstats.Stats(prof, stream=out).sort_stats("tottime").print_stats(28)
print("\n".join(l for l in out.getvalue().splitlines() if l.strip())[:6000])
