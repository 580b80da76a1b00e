# Round 5, first GPU call: the bf16x6 probe (error table + chain-kernel timings, stop rule), the
# new tests (bf16x6 at the exact mode's tolerances, config 3 against the reference's own fit), the
# per-kernel step timing in the three modes, then every older -m gpu test.
OUT=gpurun_out/r5a
mkdir -p $OUT
S=$(date +%s)
timeout 600 python -m tests.probe_bf16x6 --out $OUT/bf16x6_probe.json > $OUT/probe.log 2>&1; echo "probe rc=$? $(( $(date +%s) - S ))s"
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/r5a/bf16x6_probe.json"))
    print(json.dumps(d.get("stop_rule"), indent=1))
    for t in d.get("errors_vs_float64", []):
        print(t["model"], {k: (round(v["logits_max_abs_err_over_max_abs"] * 1e7, 2), round(v["worst_tensor_grad_max_abs_err_over_max_abs"] * 1e7, 2)) for k, v in t["modes"].items()}, "(x1e-7: logits, grads)")
    for t in d.get("distance_from_exact_f32_kernels", []):
        print(t["model"], {k: (round(v["logits_max_abs_diff_over_max_abs"] * 1e7, 2), round(v["worst_tensor_grad_max_abs_diff_over_max_abs"] * 1e7, 2)) for k, v in t["modes"].items()})
    for t in d.get("timings", []):
        print(t["model"], {k: (v["inference_forward_ms"], v["training_forward_ms"], v["backward_data_ms"]) for k, v in t["modes"].items()})
except Exception as e:
    print("no probe document:", e)
PY
tail -5 $OUT/probe.log
S=$(date +%s); timeout 900 python -m pytest tests/test_round5_gpu.py -q > $OUT/round5.log 2>&1; echo "round5 tests rc=$? $(( $(date +%s) - S ))s"
grep -n "passed\|failed\|^FAILED\|^ERROR" $OUT/round5.log | tail -40
timeout 300 python scripts/microbench_train_kernels.py --modes f32,bf16x3,bf16x6 > $OUT/train_kernels.json 2> $OUT/train_kernels.err; cat $OUT/train_kernels.json
S=$(date +%s); timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_round5_gpu.py > $OUT/older.log 2>&1; echo "older tests rc=$? $(( $(date +%s) - S ))s"
grep -n "passed\|failed\|^FAILED\|^ERROR" $OUT/older.log | tail -15
