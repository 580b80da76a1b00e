for v in new ws_dup2 new ws_dup2; do
if [ $v = new ]; then unset FFN_HIP_LIBRARY; else export FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_$v.so; fi
echo "== $v"; FFN_BF16_KERNELS=ws FFN_BF16_WAVES=8 python scripts/microbench_bf16_chain.py 2>&1 | tail -1
done
