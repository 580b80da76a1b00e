python -m pytest tests/test_round2_gpu.py -m gpu -q -k "bf16" 2>&1 | tail -15
python scripts/microbench_bf16.py 2>&1 | tail -30
