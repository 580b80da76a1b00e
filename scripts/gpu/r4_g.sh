mkdir -p gpurun_out/r4g
timeout 900 python -m pytest tests/test_round4_gpu.py -m gpu -q -k "two_waves or hardware or 512_wide" > gpurun_out/r4g/ws.log 2>&1; echo "rc=$?" >> gpurun_out/r4g/ws.log
grep -v "^  \|^$" gpurun_out/r4g/ws.log | tail -8
FFN_BF16_KERNELS=ring timeout 300 python scripts/microbench_bf16_chain.py 2>&1 | tail -1
for wv in 8 16 8 16; do
  FFN_BF16_KERNELS=ws FFN_BF16_WAVES=$wv timeout 300 python scripts/microbench_bf16_chain.py 2>&1 | tail -1
done
