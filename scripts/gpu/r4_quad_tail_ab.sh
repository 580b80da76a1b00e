mkdir -p gpurun_out/r4i
timeout 900 python -m pytest tests/test_round4_gpu.py tests/test_round3_gpu.py -m gpu -q -k "tail or micro_batched or two_rank or fit" > gpurun_out/r4i/tail.log 2>&1; echo "rc=$?" >> gpurun_out/r4i/tail.log
grep -v "^  \|^$" gpurun_out/r4i/tail.log | tail -10
for q in 1 0 1 0; do
FFN_TAIL_QUADS=$q python - <<'PY'
import json, os, sys, contextlib, io
sys.argv = ["bench.py"]
import torch, numpy as np
import bench
import fourier_feature_nets_amd as ffn
device = torch.device("cuda:0")
intr, poses = bench.synthetic_rig(20, 400)
cams = [ffn.CameraInfo.create("t%03d" % i, ffn.Resolution(400, 400), intr, p) for i, p in enumerate(poses)]
bounds = np.diag([2, 2, 2, 1]).astype(np.float32)
with contextlib.redirect_stdout(io.StringIO()):
    probe = ffn.RaySampler(bounds, cams, 128, device=device)
    images = bench.analytic_images(probe)
    del probe
out = bench.default_batch_leg(device, cams, images, bounds)
print("quads", os.environ["FFN_TAIL_QUADS"], {k: (v["ms_per_step"], v["per_ray_rate_vs_large_batch"]) for k, v in out.items() if isinstance(v, dict)})
PY
done
