mkdir -p gpurun_out/r4k
timeout 600 python -m pytest tests/test_round4_gpu.py -m gpu -q -k "small_ensemble" > gpurun_out/r4k/ens.log 2>&1; echo "rc=$?" >> gpurun_out/r4k/ens.log
grep -v "^  \|^$" gpurun_out/r4k/ens.log | tail -12
