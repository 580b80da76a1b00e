# A/B on one box, training kernels only: HEAD's mlp.hip (scripts/probes/variants/libffn_head.so) vs the tree's
for rep in 1 2 3; do
for v in head ""; do
  if [ -n "$v" ]; then export FFN_HIP_LIBRARY=scripts/probes/variants/libffn_$v.so; else unset FFN_HIP_LIBRARY; fi
  echo -n "${v:-new} tiny: "; timeout 200 python scripts/microbench_train_kernels.py --modes f32 --iters 6 2>&1 | tail -1
  echo -n "${v:-new} nerf: "; timeout 200 python scripts/microbench_train_kernels.py --modes f32 --model nerf --iters 4 2>&1 | tail -1
done
done
