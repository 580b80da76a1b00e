# round 4, first call: the new-width tests, then every GPU test, then A/B of the bias-overflow
# branch against round 3's library on the headline step
mkdir -p gpurun_out/r4a
python -m pytest tests/test_round4_gpu.py -m gpu -q -x > gpurun_out/r4a/round4.log 2>&1; echo "rc=$?" >> gpurun_out/r4a/round4.log
tail -15 gpurun_out/r4a/round4.log
python -m pytest tests -m gpu -q --deselect tests/test_round4_gpu.py > gpurun_out/r4a/all.log 2>&1; echo "rc=$?" >> gpurun_out/r4a/all.log
tail -8 gpurun_out/r4a/all.log
bash scripts/gpu/ab.sh 2>&1 | tail -6
