mkdir -p gpurun_out/r2b
python -m pytest tests -m gpu -q > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b/pytest.log
python - > gpurun_out/r2b/fps.log 2>&1 <<'PY'
import sys, time, io, contextlib
sys.path.insert(0, ".")
import numpy as np, torch
import fourier_feature_nets_amd as ffn
from bench import synthetic_rig
dev = torch.device("cuda:0")
torch.manual_seed(20080524)
model = ffn.PositionalFourierMLP(3, 4, 5.5).to(dev)
intr, poses = synthetic_rig(8, 400)
cams = [ffn.CameraInfo.create("c%d" % i, ffn.Resolution(400, 400), intr, p) for i, p in enumerate(poses)]
bounds = np.diag([2, 2, 2, 1]).astype(np.float32)
with contextlib.redirect_stdout(io.StringIO()):
    sampler = ffn.RaySampler(bounds, cams, 64, device=dev)
caster = ffn.Raycaster(model)
for fused in (False, True):
    caster.fused_render = fused
    caster.render_image(sampler, 0, 32768)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for f in range(8):
        caster.render_image(sampler, f, 32768)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("fused" if fused else "unfused", "render_image fps", 8 / dt)
caster.fused_render = True
torch.cuda.synchronize(); t0 = time.perf_counter()
for f in range(8):
    img = caster.render_image_device(sampler, f, 32768)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("fused device-only fps", 8 / dt)
import tempfile, os
with tempfile.TemporaryDirectory() as tmp:
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with ffn.FrameSink() as sink:
        for f in range(16):
            sink.submit(caster.render_image_device(sampler, f % 8, 32768), os.path.join(tmp, "f%d.png" % f))
    dt = time.perf_counter() - t0
    print("fused + FrameSink PNG fps", 16 / dt)
PY
tail -15 gpurun_out/r2b/pytest.log; cat gpurun_out/r2b/fps.log | tail -8
