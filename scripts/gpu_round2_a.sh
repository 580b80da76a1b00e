mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
python bench.py --steps 10 --warmup 2 > gpurun_out/r2a/bench1.json 2> gpurun_out/r2a/bench1.err; echo "rc=$?" >> gpurun_out/r2a/bench1.err
FFN_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 3 --warmup 1 --rays 16384 --no-target-shape > gpurun_out/r2a/bench2.json 2> gpurun_out/r2a/bench2.err; echo "rc=$?" >> gpurun_out/r2a/bench2.err
tail -5 gpurun_out/r2a/pytest.log; tail -c 600 gpurun_out/r2a/bench1.json; tail -3 gpurun_out/r2a/bench2.err; tail -c 400 gpurun_out/r2a/bench2.json
