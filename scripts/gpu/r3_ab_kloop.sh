# A/B on one box: HEAD's mlp.hip against the working tree's (weight loads interleaved with the MFMAs)
for rep in 1 2; do
for v in head ""; do
  if [ -n "$v" ]; then export FFN_HIP_LIBRARY=scripts/probes/variants/libffn_$v.so; else unset FFN_HIP_LIBRARY; fi
  echo -n "${v:-new} infer: "; timeout 200 python scripts/probes/bf16_forward_time.py 2>&1 | tail -1
  echo -n "${v:-new} tiny: "; timeout 200 python scripts/microbench_train_kernels.py --modes f32 --iters 6 2>&1 | tail -1
  echo -n "${v:-new} nerf: "; timeout 200 python scripts/microbench_train_kernels.py --modes f32 --model nerf --iters 4 2>&1 | tail -1
done
done
