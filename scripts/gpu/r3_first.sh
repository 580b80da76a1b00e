# round 3, first call: the whole GPU test-suite (incl. tests/test_round3_gpu.py), the default bench line,
# and the HIP-only long PSNR run of tests/psnr_parity.py
mkdir -p gpurun_out/r3a
( time python -m pytest tests -m gpu -q -x --durations=15 ) > gpurun_out/r3a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3a/pytest.log
tail -30 gpurun_out/r3a/pytest.log
( time python bench.py ) > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; echo "bench rc=$?"
tail -c 600 gpurun_out/r3a/bench.err
python - <<'PY'
import json
b = json.loads([l for l in open("gpurun_out/r3a/bench.json").read().strip().split("\n") if l.startswith("{")][-1])
print("ms/step", round(b["ms_per_step"], 3), "rays/s", round(b["value"]), "roofline", b["roofline"]["frac"])
for k, v in b["kernels"].items(): print(" ", k, v["avg_ms"], v["frac"])
PY
( time python -m tests.psnr_parity hip --steps 5000 --every 500 --out gpurun_out/r3a/psnr_5000.json ) > gpurun_out/r3a/psnr5000.log 2>&1; tail -5 gpurun_out/r3a/psnr5000.log
