import sys, time, os, io, contextlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench as B
import fourier_feature_nets_amd as ffn
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ffn.PositionalFourierMLP(3, 4, 5.5).to(dev)
intr, poses = B.synthetic_rig(20, 400)
cams = [ffn.CameraInfo.create("t%03d" % i, ffn.Resolution(400, 400), intr, p) for i, p in enumerate(poses)]
bounds = np.diag([2, 2, 2, 1]).astype(np.float32)
with contextlib.redirect_stdout(io.StringIO()):
    probe = ffn.RaySampler(bounds, cams, 128, device=dev)
    images = B.analytic_images(probe)
    ds = ffn.ImageDataset("train", images, bounds, cams, 128, True, True, anneal_start=0.2, num_anneal_steps=2000, device=dev)
eng = ffn.TrainEngine(model, 0.0, None)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
perm = torch.randperm(len(ds), device=dev)
import cProfile, pstats
EPOCH = len(sys.argv) > 2 and sys.argv[2] == "epoch"
def run(k):
    if EPOCH:
        rays, bounds = ds.epoch_ray_ids(perm[:k * R], R)
    for i in range(k):
        batch = perm[i * R:(i + 1) * R]
        eng.train_step(ds, batch, i, 5e-4, rays=rays[bounds[i]:bounds[i + 1]] if EPOCH else None)
run(5); torch.cuda.synchronize()
t0 = time.perf_counter(); run(50); torch.cuda.synchronize(); t1 = time.perf_counter()
print("ms/step", (t1 - t0) / 50 * 1e3)
pr = cProfile.Profile(); pr.enable(); run(50); torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
