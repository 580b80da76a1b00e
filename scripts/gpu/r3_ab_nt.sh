# interleaved A/B of library variants on one box: per-kernel training timings, tiny model, bench shape
for rep in 1 2; do
  for lib in stock ${VARIANTS}; do
    if [ $lib = stock ]; then unset FFN_HIP_LIBRARY; else export FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_${lib}.so; fi
    echo "$lib $rep $(python scripts/microbench_train_kernels.py ${MB_ARGS} 2>/dev/null | tail -1)"
  done
done
