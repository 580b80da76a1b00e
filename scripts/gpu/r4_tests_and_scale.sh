mkdir -p gpurun_out/r4j
S=$(date +%s); timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r4j/all.log 2>&1; echo "pytest rc=$? $(( $(date +%s) - S ))s"
tail -4 gpurun_out/r4j/all.log
FFN_BENCH_SHARE_GPU=1 RAYS=8192 STEPS=3 WARMUP=1 NS="1 2 4" bash scripts/gpu/scale.sh 2>&1 | tail -8
cp gpurun_out/scale/scale_curve.json gpurun_out/r4j/scale_curve_shared_gpu_functional.json
python -c "
import json; b=json.loads([l for l in open('gpurun_out/scale/bench_2.json').read().split('\n') if l.startswith('{')][-1]); print(b['collective']); print(b['cpu_baseline']['source'])"
