timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | tail -4
python scripts/microbench_train_kernels.py 2>/dev/null | tail -1
python scripts/microbench_train_kernels.py --model nerf --rays 32768 --samples 128 --iters 2 2>/dev/null | tail -1
for c in "24,15,12,18" "24,18,14,18" "24,13,10,18" "24,15,12,14"; do echo "nerf costs $c: $(FFN_UNIT_COST16=$c python scripts/microbench_train_kernels.py --model nerf --rays 32768 --samples 128 --iters 2 --modes bf16x3 2>/dev/null | tail -1 | sed 's/.*wgrad_units_bf16x3": \([0-9.]*\).*/\1/')"; done
