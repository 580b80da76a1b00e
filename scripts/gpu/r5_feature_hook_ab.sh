# (needs scripts/probes/ws_variants/feature_hook.patch applied: git apply, rebuild)
# A/B of FFN_WS_FEATURE_HOOK (features-only steps of the two-waves-per-SIMD chain kernels: the next
# segment's encoding features generated inside the K loops of both waves instead of one wave
# generating while the other multiplies): tiny NeRF and full NeRF, bf16x3 and bf16x6, interleaved on
# one box; then every test that touches those kernels.
OUT=gpurun_out/r5hook
mkdir -p $OUT
for rep in 1 2; do
for v in 0 1; do
  for model in tiny nerf; do
    R=65536; S=64; if [ $model = nerf ]; then R=16384; S=128; fi
    FFN_WS_FEATURE_HOOK=$v timeout 300 python scripts/microbench_train_kernels.py --model $model --rays $R --samples $S --iters 4 --modes bf16x3,bf16x6 > $OUT/mb_${model}_${v}_$rep.json 2>$OUT/mb.err
    python - <<PY
import json
d = json.loads([l for l in open("$OUT/mb_${model}_${v}_$rep.json") if l.startswith("{")][-1])
print("hook=$v rep=$rep $model", [(m, {k: x for k, x in d[m].items() if "forward" in k}) for m in ("bf16x3", "bf16x6")])
PY
  done
done
done 2>&1 | tee $OUT/ab.txt
S=$(date +%s); timeout 1200 python -m pytest tests/test_round5_gpu.py tests/test_round4_gpu.py tests/test_round3_gpu.py -q -x > $OUT/tests.log 2>&1; echo "tests rc=$? $(( $(date +%s) - S ))s"
tail -3 $OUT/tests.log
