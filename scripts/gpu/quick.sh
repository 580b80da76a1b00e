# quick loop: MLP kernel parity tests + the per-kernel timings of the bench workload
mkdir -p gpurun_out/quick
python -m pytest tests/test_kernels_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x -k "mlp or nerf or fit or step or wide" > gpurun_out/quick/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/quick/pytest.log
tail -3 gpurun_out/quick/pytest.log
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-config3 --no-config5 --no-skip-leg --no-bf16-leg ${BENCH_EXTRA} > gpurun_out/quick/bench.json 2> gpurun_out/quick/bench.err
python - <<'PY'
import json
b = json.loads(open("gpurun_out/quick/bench.json").read().strip().split("\n")[-1])
print("ms/step", round(b["ms_per_step"], 3), "rays/s", round(b["value"]))
for k, v in b["kernels"].items(): print(" ", k, v["avg_ms"], v["frac"])
if b.get("render"): print("  render fps", b["render"]["kernels_only_fps"], b["render"]["value"])
if b.get("north_star_shape"):
    n = b["north_star_shape"]; print("  north star", n["mlp_ms"], n["frac"], {k: (v["avg_ms"], v["frac"]) for k, v in n["kernels"].items()})
PY
