# exact-f32 vs opt-in split-bf16 training at convergence: 10 seeds x 5000 steps of the ensemble
# protocol through Raycaster.fit, both modes (HIP only; the reference takes 2 h per such run)
mkdir -p gpurun_out/r4L
for prec in f32 bf16x3; do
  S=$(date +%s)
  python -m tests.psnr_ensemble hip --seeds 10 --steps 5000 --report-interval 1000 --precision $prec --out gpurun_out/r4L/psnr_long_$prec.json > gpurun_out/r4L/long_$prec.log 2>&1
  echo "long ensemble $prec rc=$? $(( $(date +%s) - S ))s"
  grep "^seed" gpurun_out/r4L/long_$prec.log | head -10
done
python - <<'PY'
import json
a = json.load(open("gpurun_out/r4L/psnr_long_f32.json"))["final_val_psnr"]
b = json.load(open("gpurun_out/r4L/psnr_long_bf16x3.json"))["final_val_psnr"]
print("f32   %.3f +- %.3f (std %.3f)" % (a["mean"], a["stderr"], a["std"]))
print("bf16x3 %.3f +- %.3f (std %.3f)" % (b["mean"], b["stderr"], b["std"]))
PY
