"""Diagnostic: per-tensor gradient error of the split-bf16 training kernels against the exact-f32
kernels on the golden inputs, for ragged (257) and aligned (256) sample counts."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.test_kernels_gpu import _load_fourier, _load_nerf
g = np.load("tests/golden/models.npz")
dev = torch.device("cuda:0")
for name in ["positional", "nerf", "gaussian"]:
    for n in (257, 256, 1000):
        grads = {}
        for mode in ("f32", "bf16x3"):
            if name.startswith("nerf"):
                model, _ = _load_nerf(g, name, [4], True)
            else:
                model, _ = _load_fourier(g, name)
            torch.manual_seed(5)
            if n <= 257:
                x = torch.from_numpy(g["x"])[:n].to(dev).contiguous()
                v = torch.from_numpy(g["v"])[:n].to(dev).contiguous()
            else:
                x = torch.rand(n, 3, device=dev) * 2 - 1
                v = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=1)
            model.train_precision = mode
            y = model(x, v) if name.startswith("nerf") else model(x)
            probe = torch.linspace(-1, 1, y.numel()).reshape(y.shape).to(dev)
            (y * probe).sum().backward()
            grads[mode] = {k: p.grad.detach().double().cpu() for k, p in model.named_parameters() if p.grad is not None}
        print(name, "n =", n)
        for k in grads["f32"]:
            a, b = grads["f32"][k], grads["bf16x3"][k]
            err = (a - b).abs()
            scale = float(a.abs().max())
            big = int((err > 1e-3 * scale).sum())
            print("   %-22s scale %.2e  max/scale %.2e  median/scale %.2e  entries > 1e-3 scale: %d of %d"
                  % (k, scale, float(err.max()) / scale, float(err.median()) / scale, big, err.numel()), flush=True)
