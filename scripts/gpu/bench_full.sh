mkdir -p gpurun_out/full
S=$(date +%s); python bench.py > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err; echo rc=$? wall=$(( $(date +%s) - S ))s

python - <<'PY'
import json
b = json.loads(open("gpurun_out/full/bench.json").read().strip().split("\n")[-1])
print("ms/step", b["ms_per_step"], "value", b["value"])
for k in ("render", "collective", "north_star_shape", "config3_step", "config4_step", "empty_space_skipping", "split_bf16_inference", "split_bf16_training", "config5_step", "cpu_baseline"):
    v = b.get(k)
    if isinstance(v, dict):
        v = {a: (c if not isinstance(c, dict) else "{...}") for a, c in v.items() if a not in ("workload", "label", "path", "includes", "sample", "metric")}
    print(k, v)
c5 = b["config5_step"]
print(c5["full"].get("kernels"))
PY
