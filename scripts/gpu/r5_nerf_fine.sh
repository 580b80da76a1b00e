# Config-3 protocol with a report every 10 steps over the first 100 (how long a full-NeRF trajectory
# stays with its seed twin): HIP halves against tests/golden/psnr_ensemble_reference_nerf_fine.json
OUT=gpurun_out/r5fine
mkdir -p $OUT
for mode in f32 bf16x6; do
  suffix=""; if [ $mode != f32 ]; then suffix="_$mode"; fi
  FFN_PRECISION=$mode timeout 1500 python -m tests.psnr_ensemble hip --model nerf --opacity voxels --size 128 --cameras 20 --val-cameras 4 \
      --samples 128 --rays 1024 --steps 100 --crop-steps 1000 --report-interval 10 --anneal-steps 150 --seeds ${HIP_SEEDS:-8} --host-noise \
      --precision $mode --reference tests/golden/psnr_ensemble_reference_nerf_fine.json --out $OUT/r05_psnr_ensemble_nerf_fine$suffix.json > $OUT/ens$suffix.log 2>&1
  echo "nerf fine $mode rc=$?"
done
python - <<'PY'
import json
for name in ("nerf_fine", "nerf_fine_bf16x6"):
    d = json.load(open("gpurun_out/r5fine/r05_psnr_ensemble_%s.json" % name))
    a = d["against_reference"]
    print(name, "protocol_matches", a["protocol_matches"], [(r["step"], r["seeds"], round(r["max_abs_delta_db"], 4)) for r in a["paired_val_psnr_by_report"]])
PY
