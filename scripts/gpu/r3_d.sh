python -m pytest tests/test_kernels_gpu.py tests/test_round2_gpu.py tests/test_round3_gpu.py -m gpu -q -x -k "mlp or bf16 or nerf or step or fit" 2>&1 | tail -4
for rep in 1 2; do
echo "default $(python scripts/microbench_train_kernels.py 2>/dev/null | tail -1)"
echo "elide-none $(FFN_ELIDE_HEAD_DZ=none python scripts/microbench_train_kernels.py 2>/dev/null | tail -1)"
echo "elide-both $(FFN_ELIDE_HEAD_DZ=f32,bf16x3 python scripts/microbench_train_kernels.py 2>/dev/null | tail -1)"
done
