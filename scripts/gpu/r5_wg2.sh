OUT=gpurun_out/r5h
mkdir -p $OUT
timeout 300 python scripts/microbench_train_kernels.py --modes f32,bf16x6 2>/dev/null | tail -1 | tee $OUT/train_kernels_tiny.json
timeout 300 python scripts/microbench_train_kernels.py --modes f32,bf16x6 --model nerf --rays 16384 --samples 128 2>/dev/null | tail -1 | tee $OUT/train_kernels_nerf.json
timeout 600 python -m tests.probe_bf16x6 --out $OUT/probe.json --error-seeds 8 > /dev/null 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5h/probe.json"))
print(json.dumps(d.get("stop_rule")))
for layers, rows in d["error_ratio_over_seeds"]["ratios_split_over_exact"].items():
    for k, r in rows.items():
        if "6p" in k and "layers=2" in layers: print(" ", layers, k, r)
for t in d["distance_from_exact_f32_kernels"]:
    print(t["model"], t["modes"]["bf16x6_6p"])
for t in d.get("timings", []):
    print(t["model"], {k: (v["inference_forward_ms"], v["training_forward_ms"], v["backward_data_ms"]) for k, v in t["modes"].items()})
PY
S=$(date +%s); timeout 900 python -m pytest tests/test_round5_gpu.py -q > $OUT/round5.log 2>&1; echo "round5 tests rc=$? $(( $(date +%s) - S ))s"
grep -n "passed\|failed\|^FAILED\|^ERROR" $OUT/round5.log | tail -10
