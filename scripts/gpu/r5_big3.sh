OUT=gpurun_out/r5big
mkdir -p $OUT
S=$(date +%s); timeout 600 python -m pytest tests/test_round5_gpu.py -q -k "opacity_model_renders" > $OUT/own.log 2>&1; echo "test rc=$? $(( $(date +%s) - S ))s"; tail -3 $OUT/own.log
for c in 512 1024; do
  timeout 300 python scripts/microbench_train_kernels.py --modes f32 --model nerf --channels $c --rays 8192 --samples 128 2>/dev/null | tail -1
  timeout 300 python scripts/microbench_train_kernels.py --modes f32 --model mlp --channels $c --rays 16384 --samples 64 2>/dev/null | tail -1
done | tee $OUT/big_kernels.txt
