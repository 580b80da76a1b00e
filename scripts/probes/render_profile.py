import sys, os, io, time, contextlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench as B
import fourier_feature_nets_amd as ffn
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ffn.PositionalFourierMLP(3, 4, 5.5).to(dev)
intr, poses = B.synthetic_rig(8, 400)
cams = [ffn.CameraInfo.create("t%03d" % i, ffn.Resolution(400, 400), intr, p) for i, p in enumerate(poses)]
bounds = np.diag([2, 2, 2, 1]).astype(np.float32)
with contextlib.redirect_stdout(io.StringIO()):
    sampler = ffn.RaySampler(bounds, cams, 64, device=dev)
caster = ffn.Raycaster(model)
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
caster.render_image(sampler, 0, bs); torch.cuda.synchronize()
t0 = time.perf_counter()
for f in range(8):
    caster.render_image(sampler, f, bs)
torch.cuda.synchronize()
print("batch", bs, "fps", 8 / (time.perf_counter() - t0))
