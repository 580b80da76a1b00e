# HIP halves of round 5's two extra PSNR protocols against their reference fixtures: HIP_SEEDS seeds
# (default 24; the fixtures' seeds are a prefix: those are the seed-paired ones), exact-f32 kernels
# and the bf16x6 mode, jitter from the CPU generator like the reference (seed-paired trajectories)
OUT=gpurun_out/r5ens
mkdir -p $OUT
K_NERF=$(python -c "import json;print(len(json.load(open('tests/golden/psnr_ensemble_reference_nerf.json'))['runs']))")
K_SLOW=$(python -c "import json;print(len(json.load(open('tests/golden/psnr_ensemble_reference_slow.json'))['runs']))")
HIP_SEEDS=${HIP_SEEDS:-24}
if [ $K_NERF -lt $HIP_SEEDS ]; then K_NERF=$HIP_SEEDS; fi
if [ $K_SLOW -lt $HIP_SEEDS ]; then K_SLOW=$HIP_SEEDS; fi
echo "hip seeds: nerf $K_NERF, slow $K_SLOW"
for mode in f32 bf16x6; do
  suffix=""; if [ $mode != f32 ]; then suffix="_$mode"; fi
  FFN_PRECISION=$mode timeout 1500 python -m tests.psnr_ensemble hip --model nerf --opacity voxels --size 128 --cameras 20 --val-cameras 4 \
      --samples 128 --rays 1024 --steps 300 --crop-steps 1000 --report-interval 100 --anneal-steps 150 --seeds $K_NERF --host-noise \
      --precision $mode --reference tests/golden/psnr_ensemble_reference_nerf.json --out $OUT/r05_psnr_ensemble_nerf$suffix.json > $OUT/ens_nerf$suffix.log 2>&1
  echo "nerf $mode rc=$?"
  FFN_PRECISION=$mode timeout 1500 python -m tests.psnr_ensemble hip --rays 4096 --lr 1e-4 --steps 500 --crop-steps 125 --report-interval 125 \
      --anneal-steps 250 --seeds $K_SLOW --host-noise --precision $mode --reference tests/golden/psnr_ensemble_reference_slow.json \
      --out $OUT/r05_psnr_ensemble_slow$suffix.json > $OUT/ens_slow$suffix.log 2>&1
  echo "slow $mode rc=$?"
done
python - <<'PY'
import json
for name in ("nerf", "nerf_bf16x6", "slow", "slow_bf16x6"):
    try:
        d = json.load(open("gpurun_out/r5ens/r05_psnr_ensemble_%s.json" % name))
    except Exception as e:
        print(name, "missing", e); continue
    a = d["against_reference"]
    print(name, "seeds", len(d["runs"]), "protocol_matches", a["protocol_matches"], a["resolution"]["verdict"],
          "delta %.4f +- %.4f" % (a["delta_mean_db"], a["stderr_of_delta_db"]))
    print("   paired max |delta| by report:", [(r["step"], round(r["max_abs_delta_db"], 4)) for r in a["paired_val_psnr_by_report"]])
PY
