export FFN_UNIT_COST16=24,15,12,48
for lib in stock ${VARIANTS}; do
  if [ $lib = stock ]; then unset FFN_HIP_LIBRARY; else export FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_${lib}.so; fi
  echo "$lib $(python scripts/probes/wgrad_alone.py 2>/dev/null | tail -1)"
done
