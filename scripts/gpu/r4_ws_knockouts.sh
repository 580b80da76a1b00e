mkdir -p gpurun_out/r4d
timeout 900 python -m pytest tests/test_round4_gpu.py -m gpu -q > gpurun_out/r4d/round4.log 2>&1; echo "rc=$?" >> gpurun_out/r4d/round4.log
grep -v "^  \|^$" gpurun_out/r4d/round4.log | tail -8
for lib in new ws_noload ws_nox ws_nofeat ws_nomfma ws_nobar ws_noepi; do
  if [ $lib = new ]; then unset FFN_HIP_LIBRARY; else export FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_$lib.so; fi
  echo "== $lib"; timeout 300 python scripts/microbench_bf16_chain.py --models tiny 2>&1 | tail -1
done
