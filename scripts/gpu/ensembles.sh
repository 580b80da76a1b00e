# HIP halves of the PSNR ensembles against the reference's own `fit` (fixtures under tests/golden/,
# written in the build container from /root/reference by `python -m tests.psnr_ensemble reference`):
#   PROTOCOLS="tiny slow nerf nerf_fine nerf_slow"  MODES="f32 bf16x6"  ROUND=r06  HIP_SEEDS=24
# Jitter comes from the CPU generator like the reference's (seed-paired trajectories).
OUT=gpurun_out/ens
mkdir -p $OUT
ROUND=${ROUND:-r06}
NERF="--model nerf --opacity voxels --size 128 --cameras 20 --val-cameras 4 --samples 128 --crop-steps 1000 --anneal-steps 150"
for proto in ${PROTOCOLS:-nerf_slow}; do
  case $proto in
    tiny)      ARGS="";                                                                         REF=psnr_ensemble_reference.json;;
    slow)      ARGS="--rays 4096 --lr 1e-4 --steps 500 --crop-steps 125 --report-interval 125 --anneal-steps 250"; REF=psnr_ensemble_reference_slow.json;;
    nerf)      ARGS="$NERF --rays 1024 --steps 300 --report-interval 100";                      REF=psnr_ensemble_reference_nerf.json;;
    nerf_fine) ARGS="$NERF --rays 1024 --steps 100 --report-interval 10";                       REF=psnr_ensemble_reference_nerf_fine.json;;
    nerf_slow) ARGS="$NERF --rays 4096 --lr 1e-4 --steps 300 --report-interval 25";             REF=psnr_ensemble_reference_nerf_slow.json;;
  esac
  K=$(python -c "import json;print(max(${HIP_SEEDS:-0}, len(json.load(open('tests/golden/$REF'))['runs'])))")
  for mode in ${MODES:-f32 bf16x6}; do
    suffix=""; if [ $mode != f32 ]; then suffix="_$mode"; fi
    name=${ROUND}_psnr_ensemble_$proto$suffix
    FFN_PRECISION=$mode timeout ${TIMEOUT:-1500} python -m tests.psnr_ensemble hip $ARGS --seeds $K --host-noise --precision $mode \
        --reference tests/golden/$REF --out $OUT/$name.json > $OUT/$name.log 2>&1
    echo "$proto $mode ($K seeds) rc=$?"
    python - <<PY
import json
try:
    d = json.load(open("$OUT/$name.json"))
    a = d["against_reference"]
    print("$name", "seeds", len(d["runs"]), "protocol_matches", a["protocol_matches"], a["resolution"]["verdict"],
          "delta %.4f +- %.4f" % (a["delta_mean_db"], a["stderr_of_delta_db"]))
    print("   paired max |delta| by report:", [(r["step"], r["seeds"], round(r["max_abs_delta_db"], 4)) for r in a["paired_val_psnr_by_report"]])
except Exception as e:
    print("$name", "missing", e)
PY
  done
done
