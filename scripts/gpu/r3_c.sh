mkdir -p gpurun_out/r3c
( time python bench.py ) > gpurun_out/r3c/bench.json 2> gpurun_out/r3c/bench.err; echo "bench rc=$?"
tail -c 400 gpurun_out/r3c/bench.err
python - <<'PY'
import json
b = json.loads([l for l in open("gpurun_out/r3c/bench.json").read().strip().split("\n") if l.startswith("{")][-1])
print("ms/step", round(b["ms_per_step"], 3), "rays/s", round(b["value"]), "roofline", b["roofline"]["frac"])
for k, v in b["kernels"].items(): print(" ", k, v["avg_ms"], v["frac"])
print("render", {k: b["render"][k] for k in ("kernels_only_fps", "rays_per_frame", "roofline")})
print("default_batch", b["default_batch_step"])
print("bf16 train", b["split_bf16_training"])
PY
FFN_BENCH_SHARE_GPU=1 python bench.py --gpus 8 --rays 8192 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r3c/bench_8ranks_shared_gpu.json 2> gpurun_out/r3c/bench8.err; echo "bench8 rc=$?"; tail -c 300 gpurun_out/r3c/bench8.err
python - <<'PY'
import json
b = json.loads([l for l in open("gpurun_out/r3c/bench_8ranks_shared_gpu.json").read().strip().split("\n") if l.startswith("{")][-1])
print("8 ranks shared:", b["n_gpus"], round(b["ms_per_step"], 2), b["collective"])
PY
