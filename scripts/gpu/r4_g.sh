mkdir -p gpurun_out/r4g
timeout 900 python -m pytest tests/test_round4_gpu.py -m gpu -q -k "two_waves or hardware" > gpurun_out/r4g/ws.log 2>&1; echo "rc=$?" >> gpurun_out/r4g/ws.log
grep -v "^  \|^$" gpurun_out/r4g/ws.log | tail -8
for which in ring ws; do
  FFN_BF16_KERNELS=$which timeout 300 python scripts/microbench_bf16_chain.py 2>&1 | tail -1
done
export FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_ws_stamps.so
python scripts/probes/ws_stamps.py tiny 2>&1 | tail -4
