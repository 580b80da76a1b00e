# timing-only variants of the bf16x6 weight-gradient kernel (scripts/probes/wg6_variants), interleaved with the product
OUT=gpurun_out/r5i
mkdir -p $OUT
for lib in new wg6_nopin wg6_nofence wg6_noconv wg6_pin4 new; do
  if [ $lib = new ]; then unset FFN_HIP_LIBRARY; else export FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_$lib.so; fi
  echo "== $lib"; timeout 300 python scripts/microbench_train_kernels.py --modes bf16x6 2>/dev/null | tail -1
done 2>&1 | tee $OUT/wg6_variants.txt
