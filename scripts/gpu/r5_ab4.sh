OUT=gpurun_out/r5d
mkdir -p $OUT
for v in 2 1 2 1; do
  export FFN_BF16X6_FWD_ACCS=$v
  echo "== bf16x6 forward accumulators: $v"
  timeout 300 python scripts/microbench_train_kernels.py --modes bf16x6 2>/dev/null | tail -1
done 2>&1 | tee $OUT/ab_fwd_accs.txt
for v in 2 1; do
  export FFN_BF16X6_FWD_ACCS=$v
  timeout 600 python -m tests.probe_bf16x6 --out $OUT/probe_accs$v.json --error-seeds 8 --skip-timing > /dev/null 2>&1
  python - <<PY
import json
d = json.load(open("gpurun_out/r5d/probe_accs$v.json"))
print("forward accumulators $v")
for layers, rows in d["error_ratio_over_seeds"]["ratios_split_over_exact"].items():
    for k, r in rows.items():
        if "6p" in k: print(" ", layers, k, r)
t = d["errors_vs_float64"][0]["modes"]
for k in ("f32", "bf16x6_6p"):
    print(" ", k, {a: [round(x * 1e7, 2) for x in b] for a, b in t[k]["grad_err_per_tensor_max_rms"].items()})
PY
done 2>&1 | tee -a $OUT/ab_fwd_accs.txt
