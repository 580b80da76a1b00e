OUT=gpurun_out/r5big
mkdir -p $OUT
S=$(date +%s); timeout 1200 python -m pytest tests/test_round4_gpu.py tests/test_round2_gpu.py -q -k "any_layer_width or focus" > $OUT/widths.log 2>&1; echo "tests rc=$? $(( $(date +%s) - S ))s"
grep -n "passed\|failed\|^FAILED\|^ERROR" $OUT/widths.log | tail -20
grep -n "^E " $OUT/widths.log | grep -v "where\|+  " | head -20
