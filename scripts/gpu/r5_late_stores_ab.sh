# (needs scripts/probes/ws_variants/late_stores.patch applied: git apply, rebuild)
# A/B of FFN_WS_LATE_STORES (the younger wave of every SIMD issues its slab stores behind the step's
# last barrier): training kernels of the tiny NeRF and the full NeRF, bf16x3 and bf16x6, interleaved
# on one box; then the tests that compare those kernels' outputs (slabs, dZ, gradients).
OUT=gpurun_out/r5late
mkdir -p $OUT
for rep in 1 2; do
for v in 0 1; do
  for model in tiny nerf; do
    R=65536; S=64; if [ $model = nerf ]; then R=16384; S=128; fi
    FFN_WS_LATE_STORES=$v timeout 300 python scripts/microbench_train_kernels.py --model $model --rays $R --samples $S --iters 4 --modes bf16x3,bf16x6 > $OUT/mb_${model}_${v}_$rep.json 2>$OUT/mb.err
    python - <<PY
import json
d = json.loads([l for l in open("$OUT/mb_${model}_${v}_$rep.json") if l.startswith("{")][-1])
print("late=$v rep=$rep $model", [(m, d[m]) for m in ("bf16x3", "bf16x6")])
PY
  done
done
done 2>&1 | tee $OUT/ab.txt
S=$(date +%s); timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_round4_gpu.py -q -x > $OUT/tests.log 2>&1; echo "tests rc=$? $(( $(date +%s) - S ))s"
tail -3 $OUT/tests.log
