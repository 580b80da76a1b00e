# 512-wide fused render: config-5 leg of the bench alone + the narrow render leg for regression
set -e
mkdir -p gpurun_out
python - <<'PY' > gpurun_out/wide_render.json
import json, sys, torch
sys.argv = ["bench.py"]
import bench
import numpy as np
dev = torch.device("cuda:0")
bounds = np.diag([2, 2, 2, 1]).astype(np.float32)
out = bench.config5_leg(dev, bounds, steps=2)
print(json.dumps(out, indent=1))
PY
cat gpurun_out/wide_render.json | tail -25
