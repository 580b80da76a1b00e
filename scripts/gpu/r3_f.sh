python -m pytest tests -m gpu -q -x 2>&1 | tail -5
python scripts/probes/default_batch.py tiny 200 2>&1 | tail -1
FFN_TAIL_PAIRS=0 python scripts/probes/default_batch.py tiny 200 2>&1 | tail -1
python scripts/probes/default_batch.py nerf 100 2>&1 | tail -1
FFN_TAIL_PAIRS=0 python scripts/probes/default_batch.py nerf 100 2>&1 | tail -1
