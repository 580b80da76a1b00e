# every -m gpu test in an opt-in arithmetic mode (default bf16x6): PRECISION=bf16x6 bash scripts/gpu/tests_precision.sh
MODE=${PRECISION:-bf16x6}
OUT=gpurun_out/t_$MODE
mkdir -p $OUT
S=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -q --precision $MODE -rs > $OUT/pytest.log 2>&1; echo "pytest --precision $MODE rc=$? $(( $(date +%s) - S ))s" | tee -a $OUT/pytest.log
grep -n "passed\|failed\|^FAILED\|^ERROR" $OUT/pytest.log | tail -80
