mkdir -p gpurun_out/r3b
( time python -m pytest tests -m gpu -q -x ) > gpurun_out/r3b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3b/pytest.log
grep -n "passed\|failed\|FAILED\|ERROR\|rc=" gpurun_out/r3b/pytest.log | tail -8
python scripts/probes/default_batch.py tiny 200 2>&1 | tail -1
python scripts/probes/default_batch.py nerf 100 2>&1 | tail -1
python -m tests.psnr_parity hip --oracle profiles/r03_psnr_parity_oracle.json --out gpurun_out/r3b/psnr_parity_400.json 2>&1 | tail -2
