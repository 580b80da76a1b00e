# Build container: copies the reference halves computed so far (tests/golden/_psnr_oracle_config3/, written by
# `python -m tests.psnr_ensemble reference ... --resume`, see tests/psnr_ensemble.py) into the committed fixtures
set -e
D=tests/golden/_psnr_oracle_config3
python - <<'PY'
import json
for src, dst in (("nerf24.json", "psnr_ensemble_reference_nerf.json"), ("nerf_slow.json", "psnr_ensemble_reference_nerf_slow.json")):
    d = json.load(open("tests/golden/_psnr_oracle_config3/" + src))
    json.dump(d, open("tests/golden/" + dst, "w"), indent=1)
    print(dst, len(d["runs"]), "runs of", len(d["protocol"]["seeds"]), "complete" if d["complete"] else "incomplete",
          d["final_val_psnr"]["mean"], d["final_val_psnr"]["stderr"])
PY
