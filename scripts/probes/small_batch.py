import contextlib, io, sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
import fourier_feature_nets_amd as ffn
dev = torch.device("cuda:0")
intr, poses = bench.synthetic_rig(20, 400)
cams = [ffn.CameraInfo.create("c%d" % i, ffn.Resolution(400, 400), intr, p) for i, p in enumerate(poses)]
bounds = np.diag([2, 2, 2, 1]).astype(np.float32)
with contextlib.redirect_stdout(io.StringIO()):
    probe = ffn.RaySampler(bounds, cams, 128, device=dev)
    images = bench.analytic_images(probe)
    del probe
    ds = ffn.ImageDataset("train", images, bounds, cams, 128, True, True, anneal_start=0.2, num_anneal_steps=2000, device=dev)
for name, model in (("nerf", ffn.NeRF(8, 256, 9, 10, 3, 4, [4], True).to(dev)), ("tiny", ffn.PositionalFourierMLP(3, 4, 5.5).to(dev))):
    for prec in ("f32", "bf16x3"):
        model.train_precision = prec
        engine = ffn.TrainEngine(model, 0.0, None)
        valid = torch.nonzero(ds.sampler.valid != 0).flatten()
        gen = torch.Generator(device=dev).manual_seed(1)
        def step(i):
            pick = torch.randint(0, valid.numel(), (1024,), device=dev, generator=gen)
            return engine.train_step(ds, valid[pick], i, 5e-4)
        for i in range(5): step(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(5, 55): step(i)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        print(name, prec, "batch 1024 x 128: %.3f ms/step  %.0f rays/s" % (dt * 1e3, 1024 / dt))
