# every -m gpu test (round 5 first), summaries into gpurun_out/r5t/
OUT=gpurun_out/r5t
mkdir -p $OUT
S=$(date +%s); timeout 900 python -m pytest tests/test_round5_gpu.py -q > $OUT/round5.log 2>&1; echo "round5 tests rc=$? $(( $(date +%s) - S ))s"
grep -n "passed\|failed\|^FAILED\|^ERROR" $OUT/round5.log | tail -40
S=$(date +%s); timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_round5_gpu.py > $OUT/older.log 2>&1; echo "older tests rc=$? $(( $(date +%s) - S ))s"
grep -n "passed\|failed\|^FAILED\|^ERROR" $OUT/older.log | tail -15
timeout 300 python scripts/microbench_train_kernels.py --modes f32,bf16x3,bf16x6 2>/dev/null | tee $OUT/train_kernels.json
