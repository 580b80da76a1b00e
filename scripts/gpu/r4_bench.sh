mkdir -p gpurun_out/r4bench
S=$(date +%s); python bench.py > gpurun_out/r4bench/bench.json 2> gpurun_out/r4bench/bench.err; echo "bench rc=$? wall=$(( $(date +%s) - S ))s"
python - <<'PY'
import json
b = json.loads([l for l in open("gpurun_out/r4bench/bench.json").read().strip().split("\n") if l.startswith("{")][-1])
print("ms/step", b["ms_per_step"], "value", b["value"], b["roofline"]["frac"], b["roofline"]["traffic"], b["config"]["commit"])
print(b["default_batch_step"]["tiny"]["per_ray_rate_vs_large_batch"], b["default_batch_step"]["nerf"]["per_ray_rate_vs_large_batch"])
PY
