# HIP half of the PSNR ensemble in the f32-accurate split mode (the reference's 24 seeds first, 100 in all)
OUT=gpurun_out/r5ens
mkdir -p $OUT
S=$(date +%s)
FFN_PRECISION=bf16x6 timeout 1500 python -m tests.psnr_ensemble hip --reference tests/golden/psnr_ensemble_reference.json --seeds 100 --precision bf16x6 --out $OUT/r05_psnr_ensemble_bf16x6.json > $OUT/ens_x6.log 2>&1
echo "x6 ensemble rc=$? $(( $(date +%s) - S ))s"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5ens/r05_psnr_ensemble_bf16x6.json"))
a = d["against_reference"]
print(json.dumps(a["resolution"], indent=1))
print({k: a[k] for k in ("delta_mean_db", "stderr_of_delta_db", "verdict")}, d["final_val_psnr"]["mean"], d["final_val_psnr"]["stderr"])
print([ (r["step"], round(r["mean_delta_db"], 4), round(r["max_abs_delta_db"], 4)) for r in a["paired_val_psnr_by_report"]])
PY
