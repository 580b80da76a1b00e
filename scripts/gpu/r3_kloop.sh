# knock-outs of the f32 K loop: inference forward of the tiny model and the full NeRF, 2^22 samples
for v in "" kloop_noload kloop_neither; do
  if [ -n "$v" ]; then export FFN_HIP_LIBRARY=scripts/probes/variants/libffn_$v.so; fi
  echo -n "${v:-stock}: "; timeout 200 python scripts/probes/bf16_forward_time.py 2>&1 | tail -1
done
