# calibration of the logits-head unit's cost after the 4x4x1-MFMA rewrite (tiny model and full NeRF)
for h in 6 5 4 3 2; do
  echo -n "f32 head $h tiny: "; FFN_UNIT_COST=24,13,8,$h timeout 100 python scripts/microbench_train_kernels.py --modes f32 --iters 6 2>&1 | tail -1 | sed 's/.*wgrad_units": \([0-9.]*\).*/\1/'
  echo -n "f32 head $h nerf: "; FFN_UNIT_COST=24,13,8,$h timeout 100 python scripts/microbench_train_kernels.py --modes f32 --model nerf --iters 4 2>&1 | tail -1 | sed 's/.*wgrad_units": \([0-9.]*\).*/\1/'
done
for h in 18 14 12 10 8 6; do
  echo -n "bf16 head $h tiny: "; FFN_UNIT_COST16=24,24,20,$h timeout 100 python scripts/microbench_train_kernels.py --modes bf16x3 --iters 6 2>&1 | tail -1 | sed 's/.*wgrad_units_bf16x3": \([0-9.]*\).*/\1/'
  echo -n "bf16 head $h nerf: "; FFN_UNIT_COST16=24,24,20,$h timeout 100 python scripts/microbench_train_kernels.py --modes bf16x3 --model nerf --iters 4 2>&1 | tail -1 | sed 's/.*wgrad_units_bf16x3": \([0-9.]*\).*/\1/'
done
