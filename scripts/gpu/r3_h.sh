timeout 600 python -m pytest tests/test_round2_gpu.py tests/test_round3_gpu.py -m gpu -q -x -k "bf16" 2>&1 | grep -E "passed|failed|FAILED|Error|rror" | tail -6
for c in "24,15,12,12" "24,15,12,18" "24,15,12,24"; do echo "bf16 costs $c: $(FFN_UNIT_COST16=$c timeout 300 python scripts/microbench_train_kernels.py --modes bf16x3 2>/dev/null | tail -1 | sed 's/.*"bf16x3": //')"; done
