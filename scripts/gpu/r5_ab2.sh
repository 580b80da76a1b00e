OUT=gpurun_out/r5d
mkdir -p $OUT
for v in default train default train; do
  if [ $v = train ]; then export FFN_BF16X3_INFER=train; else unset FFN_BF16X3_INFER; fi
  echo "== bf16x3 inference instantiation: $v"
  FFN_BF16_KERNELS=ws timeout 300 python scripts/microbench_bf16_chain.py 2>&1 | tail -1
done 2>&1 | tee $OUT/ab_x3.txt
