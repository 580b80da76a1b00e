# split-bf16 weight gradients on 512-wide chains: parity test + config-5 step, off / on
mkdir -p gpurun_out/r4w
python -m pytest tests/test_round4_gpu.py -q -x -m gpu -k "512_wide" 2>&1 | tail -5
for v in 0 1 0 1; do
FFN_WIDE_WGRAD16=$v python - <<'PY' 2>&1 | tail -3
import json, os, torch, bench
bounds = torch.eye(4)
bounds[:3, :3] *= 2.0
out = bench.config5_leg(torch.device("cuda:0"), bounds, steps=3)
print("FFN_WIDE_WGRAD16=%s" % os.environ["FFN_WIDE_WGRAD16"], json.dumps({k: out[k] for k in ("full", "split_bf16_training")})[:900])
PY
done
