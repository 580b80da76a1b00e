mkdir -p gpurun_out/r4h
timeout 900 python -m pytest tests/test_round4_gpu.py -m gpu -q > gpurun_out/r4h/r4.log 2>&1; echo "rc=$?" >> gpurun_out/r4h/r4.log
grep -v "^  \|^$" gpurun_out/r4h/r4.log | tail -14
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_round4_gpu.py -x > gpurun_out/r4h/all.log 2>&1; echo "rc=$?" >> gpurun_out/r4h/all.log
tail -4 gpurun_out/r4h/all.log
