# the whole GPU test-suite, no -x, durations
mkdir -p gpurun_out/r3t
( time python -m pytest tests -m gpu -q --durations=12 ) > gpurun_out/r3t/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3t/pytest.log
grep -n "passed\|failed\|FAILED\|ERROR\|rc=" gpurun_out/r3t/pytest.log | tail -30
