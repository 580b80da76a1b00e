"""Times the split-bf16 inference forward (ffn_mlp_forward_bf16x3) of the tiny model and the full
NeRF on 2^22 samples:  python scripts/probes/bf16_forward_time.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fourier_feature_nets_amd as ffn  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1)
n = 1 << 22
x = torch.rand(n, 3, device=dev) * 2 - 1
v = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=1)
out = {}
for name, model, views in (("tiny", ffn.PositionalFourierMLP(3, 4, 5.5).to(dev), None),
                           ("nerf", ffn.NeRF(8, 256, 9, 10, 3, 4, [4], True).to(dev), v)):
    prog = model.program()
    for mode in ("f32", "bf16x3"):
        fn = (lambda: prog.forward16(x, views)) if mode == "bf16x3" else (lambda: prog.forward(x, views, None))
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out["%s_%s_ms" % (name, mode)] = round(e0.elapsed_time(e1) / 4, 3)
print(json.dumps(out))
