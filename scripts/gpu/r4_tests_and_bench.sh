mkdir -p gpurun_out/r4k
S=$(date +%s); timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r4k/all.log 2>&1; echo "pytest rc=$? $(( $(date +%s) - S ))s"
tail -4 gpurun_out/r4k/all.log
bash scripts/gpu/r4_bench.sh
cp gpurun_out/r4bench/bench.json gpurun_out/r4k/bench.json
