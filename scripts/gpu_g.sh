mkdir -p gpurun_out/r2g
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config5 --no-config3 --no-target-shape --no-skip-leg > gpurun_out/r2g/bench.json 2> gpurun_out/r2g/bench.err; tail -2 gpurun_out/r2g/bench.err
python - <<'PY'
import json
b = json.loads(open("gpurun_out/r2g/bench.json").read().strip().split("\n")[-1])
print({k: v for k, v in b["split_bf16_inference"].items() if k != "label"})
PY
