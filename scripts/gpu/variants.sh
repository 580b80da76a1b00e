# timing-only variants of the fused MLP (results may be wrong): fwd-train / infer per variant, one box
mkdir -p gpurun_out/var
for lib in ${VARIANTS}; do
  if [ $lib = new ]; then unset FFN_HIP_LIBRARY; else export FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_$lib.so; fi
  echo "== $lib"; python scripts/microbench_mlp.py --iters 8 2>&1 | grep fwd
done
