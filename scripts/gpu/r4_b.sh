mkdir -p gpurun_out/r4b
timeout 600 python -m pytest tests/test_round4_gpu.py -m gpu -q -x > gpurun_out/r4b/round4.log 2>&1; echo "rc=$?" >> gpurun_out/r4b/round4.log
grep -v "^  \|^$" gpurun_out/r4b/round4.log | tail -25
for rep in 1 2; do for which in ring ws; do
  FFN_BF16_KERNELS=$which timeout 300 python scripts/microbench_bf16_chain.py 2>&1 | tail -2
done; done
