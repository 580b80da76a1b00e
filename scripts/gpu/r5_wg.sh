OUT=gpurun_out/r5g
mkdir -p $OUT
for v in bf16x6 f32 bf16x6 f32; do
  export FFN_BF16X6_WGRAD=$v
  echo "== bf16x6 weight-gradient units: $v"
  timeout 300 python scripts/microbench_train_kernels.py --modes bf16x6 2>/dev/null | tail -1
done 2>&1 | tee $OUT/ab_wgrad.txt
unset FFN_BF16X6_WGRAD
timeout 300 python scripts/microbench_train_kernels.py --modes f32,bf16x6 --model nerf --rays 16384 --samples 128 2>/dev/null | tail -1 | tee -a $OUT/ab_wgrad.txt
timeout 600 python -m tests.probe_bf16x6 --out $OUT/probe.json --error-seeds 8 --skip-timing > /dev/null 2>&1
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5g/probe.json"))
for layers, rows in d["error_ratio_over_seeds"]["ratios_split_over_exact"].items():
    for k, r in rows.items():
        if "6p" in k: print(" ", layers, k, r)
for t in d["errors_vs_float64"]:
    for k in ("f32", "bf16x6_6p"):
        print(" ", t["model"], k, {a: [round(x * 1e7, 2) for x in b] for a, b in t["modes"][k]["grad_err_per_tensor_max_rms"].items()})
for t in d["distance_from_exact_f32_kernels"]:
    print(t["model"], t["modes"]["bf16x6_6p"])
PY
S=$(date +%s); timeout 900 python -m pytest tests/test_round5_gpu.py -q -x > $OUT/round5.log 2>&1; echo "round5 tests rc=$? $(( $(date +%s) - S ))s"
grep -n "passed\|failed\|^FAILED\|^ERROR" $OUT/round5.log | tail -10
