for lib in ${VARIANTS}; do
  if [ $lib = new ]; then unset FFN_HIP_LIBRARY; else export FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_$lib.so; fi
  echo "== $lib"; python scripts/microbench_bf16.py 2>&1 | grep -A3 '"bf16x3"' | grep '"ms"'
done
