mkdir -p gpurun_out/t
python -m pytest tests -m gpu -q > gpurun_out/t/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t/pytest.log
grep -n "passed\|failed\|FAILED\|rc=" gpurun_out/t/pytest.log | tail -12
