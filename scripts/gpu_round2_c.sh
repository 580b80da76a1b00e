mkdir -p gpurun_out/r2c
python -m pytest tests -m gpu -q > gpurun_out/r2c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c/pytest.log
tail -4 gpurun_out/r2c/pytest.log
python bench.py --steps 10 --warmup 2 > gpurun_out/r2c/bench1.json 2> gpurun_out/r2c/bench1.err; echo "rc=$?" >> gpurun_out/r2c/bench1.err
tail -3 gpurun_out/r2c/bench1.err; tail -c 1500 gpurun_out/r2c/bench1.json
bash scripts/gpu_profile_round2.sh
