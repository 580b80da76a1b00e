export FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_ws_bwd_nosave.so
for wv in 8 16 8 16; do
  FFN_BF16_KERNELS=ws FFN_BF16_WAVES=$wv timeout 300 python scripts/microbench_bf16_chain.py 2>&1 | tail -1
done
