python scripts/probes/default_batch.py tiny 200 2>&1 | tail -1
FFN_FORCE_WIDE=1 python scripts/probes/default_batch.py tiny 200 2>&1 | tail -1
python scripts/probes/default_batch.py nerf 100 2>&1 | tail -1
FFN_FORCE_WIDE=1 python scripts/probes/default_batch.py nerf 100 2>&1 | tail -1
