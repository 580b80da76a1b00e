"""Per-step wall time of the opt-in skip training leg (is anything re-planned per step?)."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench as B
import fourier_feature_nets_amd as ffn
dev = torch.device("cuda:0")
intr, poses = B.synthetic_rig(100, 400)
cams = [ffn.CameraInfo.create("c%d" % i, ffn.Resolution(400, 400), intr, p) for i, p in enumerate(poses)]
bounds = np.diag([2, 2, 2, 1]).astype(np.float32)
with contextlib.redirect_stdout(io.StringIO()):
    probe = ffn.RaySampler(bounds, cams, 64, device=dev)
    images = B.analytic_images(probe)
    del probe
    ds = ffn.ImageDataset("train", images, bounds, cams, 64, True, True, anneal_start=0.2, num_anneal_steps=2000, device=dev)
torch.manual_seed(20080524)
model = ffn.PositionalFourierMLP(3, 4, 5.5).to(dev)
centres = ffn.OccupancyGrid.cell_centres(bounds, 128, dev)
logits = torch.zeros((centres.shape[0], 4), device=dev)
logits[:, 3] = torch.where(centres.norm(dim=1) < 0.6, 10.0, -30.0)
grid = ffn.OccupancyGrid.from_logits(logits, bounds, 128, 0.01, True)
engine = ffn.TrainEngine(model, 0.0, None)
engine.occupancy = grid
valid = torch.nonzero(ds.sampler.valid != 0).flatten()
gen = torch.Generator(device=dev).manual_seed(99)
prog = model.program()
for mode in ("f32", "bf16x3"):
    model.train_precision = mode
    for step in range(8):
        pick = torch.randint(0, valid.numel(), (65536,), device=dev, generator=gen)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        engine.train_step(ds, valid[pick], step, 5e-4)
        torch.cuda.synchronize()
        print(mode, step, "%.2f ms" % (1e3 * (time.perf_counter() - t0)), "workspaces", sorted(prog._workspaces.keys()))
