mkdir -p gpurun_out/r6i
for rep in 1 2; do
  for v in r5base new; do
    if [ $v = new ]; then unset FFN_HIP_LIBRARY; else export FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_$v.so; fi
    echo "== $v rep $rep"
    timeout 300 python scripts/microbench_train_kernels.py --model tiny --modes bf16x6 --iters 4 2>&1 | tail -1
    timeout 300 python scripts/microbench_train_kernels.py --model nerf --rays 16384 --samples 128 --modes bf16x6 --iters 3 2>&1 | tail -1
  done
done | tee gpurun_out/r6i/ab.txt
unset FFN_HIP_LIBRARY
timeout 900 python -m pytest tests/test_round5_gpu.py tests/test_round4_gpu.py -q -x > gpurun_out/r6i/t.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r6i/t.log
PRECISION=bf16x6 timeout 300 python -m pytest tests/test_round5_gpu.py -q -x --precision bf16x6 2>&1 | tail -3
