OUT=gpurun_out/r5c
mkdir -p $OUT
timeout 900 python -m tests.probe_bf16x6 --out $OUT/bf16x6_probe.json --error-seeds 8 > $OUT/probe.log 2>&1; echo "probe rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5c/bf16x6_probe.json"))
print(json.dumps(d.get("stop_rule"), indent=1))
for layers, rows in d["error_ratio_over_seeds"]["ratios_split_over_exact"].items():
    for k, v in rows.items():
        print(layers, k, v)
for t in d["errors_vs_float64"][:1]:
    for k, v in t["modes"].items():
        print(k, {a: [round(x * 1e7, 2) for x in b] for a, b in v["grad_err_per_tensor_max_rms"].items()})
for t in d.get("timings", []):
    print(t["model"], {k: (v["inference_forward_ms"], v["training_forward_ms"], v["backward_data_ms"]) for k, v in t["modes"].items()})
PY
