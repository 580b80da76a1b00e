mkdir -p gpurun_out/r2e
python -m pytest tests/test_round2_gpu.py tests/test_pipeline_gpu.py tests/test_kernels_gpu.py -m gpu -q -k "focus or live" 2>&1 | tail -6
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config5 --no-target-shape --no-skip-leg > gpurun_out/r2e/bench.json 2> gpurun_out/r2e/bench.err; tail -2 gpurun_out/r2e/bench.err
python - <<'PY'
import json
b = json.loads(open("gpurun_out/r2e/bench.json").read().strip().split("\n")[-1])
c = b["config3_step"]; print({k: v for k, v in c.items() if k not in ("workload", "kernels")})
print(b["render"]["kernels_only_fps"], b["render"]["value"], b["render"]["with_async_d2h_and_png_fps"])
PY
