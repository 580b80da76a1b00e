"""Start-up costs of the sampler at the bench scale (100 cameras x 400x400 = 16 M rays):
ray generation + slab test (a1-a3), and the opacity-guided focus tables (a5) built with the
fused MLP as the coarse model."""
import contextlib, io, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B
import fourier_feature_nets_amd as ffn
dev = torch.device("cuda:0")
intr, poses = B.synthetic_rig(100, 400)
cams = [ffn.CameraInfo.create("t%03d" % i, ffn.Resolution(400, 400), intr, p) for i, p in enumerate(poses)]
bounds = np.diag([2, 2, 2, 1]).astype(np.float32)
quiet = io.StringIO()
for S, with_opacity in ((64, False), (128, True)):
    torch.manual_seed(0)
    coarse = ffn.PositionalFourierMLP(3, 4, 5.5).to(dev) if with_opacity else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(quiet):
        sampler = ffn.RaySampler(bounds, cams, S, True, coarse, 65536, device=dev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("RaySampler(100 x 400x400, S=%d, opacity_model=%s): %.2f s, %d rays, cdfs %s"
          % (S, "tiny NeRF" if with_opacity else None, dt, sampler.num_rays,
             None if getattr(sampler, "cdfs", None) is None else tuple(sampler.cdfs.shape)))
    rays = sampler.valid.nonzero().reshape(-1)[:65536].contiguous()
    for _ in range(3):
        sampler.sample(rays, 100)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        sampler.sample(rays, 100)
    torch.cuda.synchronize()
    print("  sample(65536 rays): %.3f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
    del sampler
