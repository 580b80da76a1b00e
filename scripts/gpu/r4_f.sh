mkdir -p gpurun_out/r4f
timeout 900 python -m pytest tests/test_round4_gpu.py -m gpu -q -k "two_waves" > gpurun_out/r4f/ws.log 2>&1; echo "rc=$?" >> gpurun_out/r4f/ws.log
grep -v "^  \|^$" gpurun_out/r4f/ws.log | tail -12
for which in ring ws; do
  FFN_BF16_KERNELS=$which timeout 300 python scripts/microbench_bf16_chain.py 2>&1 | tail -1
done
