# split-bf16 two-waves-per-SIMD kernels: what their slab stores cost, and whether ordinary stores
# (L2 write-back) absorb the epilogue bursts better than non-temporal ones; interleaved, one box
mkdir -p gpurun_out/r4s
for rep in 0 1; do
for lib in product ${VARIANTS:-ws_temporal ws_fwd_nosave}; do
  if [ $lib = product ]; then unset FFN_HIP_LIBRARY; else export FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_$lib.so; fi
  echo "== $lib rep $rep"
  python scripts/microbench_bf16_chain.py 2>&1 | tail -1
done
done
