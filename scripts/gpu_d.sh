mkdir -p gpurun_out/r2d
python -m pytest tests/test_round2_gpu.py tests/test_pipeline_gpu.py -m gpu -q -k "skipping or live or occupancy" 2>&1 | tail -5
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-config3 --no-target-shape > gpurun_out/r2d/bench.json 2> gpurun_out/r2d/bench.err; tail -2 gpurun_out/r2d/bench.err
python - <<'PY'
import json
b = json.loads(open("gpurun_out/r2d/bench.json").read().strip().split("\n")[-1])
print(b["ms_per_step"], b["render"]["kernels_only_fps"]); print(b["empty_space_skipping"])
PY
