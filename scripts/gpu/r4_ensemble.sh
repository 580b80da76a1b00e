# HIP half of the PSNR ensemble after the trainval camera-order fix: 40 seeds x 1000 steps through
# Raycaster.fit in the exact-f32 mode and in the opt-in split-bf16 mode (the first 5 seeds are
# the ones the reference half ran; ~2.5 s of training per seed + dataset start-up)
mkdir -p gpurun_out/r4e
for prec in f32 bf16x3; do
  S=$(date +%s)
  python -m tests.psnr_ensemble hip --seeds 40 --precision $prec --out gpurun_out/r4e/psnr_ensemble_hip_$prec.json > gpurun_out/r4e/ensemble_$prec.log 2>&1
  echo "ensemble $prec rc=$? $(( $(date +%s) - S ))s"
  grep "^seed" gpurun_out/r4e/ensemble_$prec.log | head -8
done
python -m pytest tests/test_round4_gpu.py tests/test_pipeline_gpu.py tests/test_round3_gpu.py -q -x -m gpu -k "ensemble or fit" 2>&1 | tail -3
