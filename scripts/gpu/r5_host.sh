OUT=gpurun_out/r5f
mkdir -p $OUT
timeout 600 python scripts/probes/host_profile.py tiny 2>&1 | tee $OUT/host_profile_tiny.txt | head -60
S=$(date +%s); timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_round4_gpu.py -q > $OUT/tests.log 2>&1; echo "round4+5 tests rc=$? $(( $(date +%s) - S ))s"
grep -n "passed\|failed\|^FAILED\|^ERROR" $OUT/tests.log | tail -20
