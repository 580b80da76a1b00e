# Round 5: the new tests, then single bench legs (default batch with the round-5 protocol, the
# f32_accurate_split block) without the whole bench line
OUT=gpurun_out/r5e
mkdir -p $OUT
S=$(date +%s); timeout 900 python -m pytest tests/test_round5_gpu.py -q > $OUT/round5.log 2>&1; echo "round5 tests rc=$? $(( $(date +%s) - S ))s"
grep -n "passed\|failed\|^FAILED\|^ERROR" $OUT/round5.log | tail -20
timeout 900 python - > $OUT/legs.json 2> $OUT/legs.err <<'PY'
import contextlib, io, json, sys
import numpy as np, torch
sys.argv = ["bench.py"]
import bench
import fourier_feature_nets_amd as ffn
device = torch.device("cuda", 0)
intr, poses = bench.synthetic_rig(100, 400)
cams = [ffn.CameraInfo.create("train%03d" % i, ffn.Resolution(400, 400), intr, p) for i, p in enumerate(poses)]
bounds = np.diag([2, 2, 2, 1]).astype(np.float32)
with contextlib.redirect_stdout(io.StringIO()):
    probe = ffn.RaySampler(bounds, cams, 64, device=device)
    images = bench.analytic_images(probe)
    del probe
    dataset = ffn.ImageDataset("train", images, bounds, cams, 64, True, True, anneal_start=0.2, num_anneal_steps=2000, device=device)
out = {}
out["f32_accurate_split"] = bench.bf16_train_leg(device, dataset, 65536, 64, mode="bf16x6")
out["split_bf16_training"] = bench.bf16_train_leg(device, dataset, 65536, 64, mode="bf16x3")
del dataset
torch.cuda.empty_cache()
out["default_batch_step"] = bench.default_batch_leg(device, cams, images, bounds)
print(json.dumps(out, indent=1))
PY
echo "legs rc=$?"; tail -3 $OUT/legs.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5e/legs.json"))
for k in ("f32_accurate_split", "split_bf16_training"):
    v = d[k]
    print(k, v["train_step_ms_interleaved_runs"], v["speedup_vs_exact_f32_step"], v["kernels"], "grad rel l2", v["first_step_gradient_relative_l2_error"])
print(json.dumps(d["default_batch_step"], indent=1))
PY
