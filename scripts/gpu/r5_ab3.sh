OUT=gpurun_out/r5d
mkdir -p $OUT
for v in 8 16 8 16; do
  export FFN_BF16X6_WAVES=$v
  echo "== bf16x6 waves: $v"
  timeout 300 python scripts/microbench_train_kernels.py --modes bf16x6 2>/dev/null | tail -1
  timeout 300 python scripts/microbench_train_kernels.py --modes bf16x6 --model nerf --rays 16384 --samples 128 2>/dev/null | tail -1
done 2>&1 | tee $OUT/ab_waves.txt
