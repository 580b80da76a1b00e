for lib in stock wg16_nc_nodep wg16_nc_noreq wg16_nc_nodep_noreq; do
  if [ $lib = stock ]; then unset FFN_HIP_LIBRARY; else export FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_${lib}.so; fi
  echo "$lib $(python scripts/probes/wgrad_alone.py 2>/dev/null | tail -1)"
done
