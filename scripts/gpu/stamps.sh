# Round 5: phase timeline (cycle stamps) of the bf16x6 / bf16x3 forward kernels, the probe with the
# seed spread of the error ratios, the round-5 tests
OUT=gpurun_out/r5b
mkdir -p $OUT
export FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_ws_stamps.so
for m in "tiny bf16x6" "tiny bf16x6 train" "tiny bf16x3" "mlp8 bf16x6" "nerf bf16x6"; do
  timeout 200 python scripts/probes/ws_stamps.py $m 2>&1 | tail -5
done | tee $OUT/stamps.txt
unset FFN_HIP_LIBRARY
timeout 600 python -m tests.probe_bf16x6 --out $OUT/bf16x6_probe.json --error-seeds 8 --skip-timing > $OUT/probe.log 2>&1; echo "probe rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5b/bf16x6_probe.json"))
print(json.dumps(d.get("error_ratio_over_seeds"), indent=1))
for t in d["errors_vs_float64"][:1]:
    for k, v in t["modes"].items():
        print(k, {a: [round(x * 1e7, 2) for x in b] for a, b in v["grad_err_per_tensor_max_rms"].items()})
PY
S=$(date +%s); timeout 900 python -m pytest tests/test_round5_gpu.py -q > $OUT/round5.log 2>&1; echo "round5 tests rc=$? $(( $(date +%s) - S ))s"
grep -n "passed\|failed\|^FAILED\|^ERROR" $OUT/round5.log | tail -40
