# calibration sweep of the folded-unit costs (full NeRF, 65536 x 64 samples, f32 weight gradients)
for fc in 7,5,4,3 7,5,4,4 7,5,4,5 8,5,4,3 8,5,4,4 8,5,4,5 9,5,4,4 6,5,4,4 7,5,4,3; do
  echo -n "fold $fc: "; FFN_FOLD_COST=$fc timeout 300 python scripts/microbench_train_kernels.py --model nerf --samples 64 --modes f32 --iters 6 2>&1 | tail -1 | sed 's/.*wgrad_units": \([0-9.]*\).*/\1/'
done
