"""Diagnostic: gradient errors of the exact-f32 and split-bf16 training kernels against the
reference goldens (tests/golden/models.npz), per model: worst per-tensor max error relative to the
tensor's scale, relative L2 over all compared entries, Sigma|g| error."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.test_kernels_gpu import _load_fourier, _load_nerf
g = np.load("tests/golden/models.npz")
dev = torch.device("cuda:0")
for name in ["mlp", "basic", "positional", "gaussian", "nerf", "nerf_small"]:
    for mode in ("f32", "bf16x3"):
        if name.startswith("nerf"):
            model, _ = _load_nerf(g, name, [4] if name == "nerf" else [2], name == "nerf")
            args = (torch.from_numpy(g["x"]).to(dev), torch.from_numpy(g["v"]).to(dev))
        else:
            model, _ = _load_fourier(g, name)
            args = (torch.from_numpy(g["x"]).to(dev),)
        model.train_precision = mode
        y = model(*args)
        probe = torch.linspace(-1, 1, y.numel()).reshape(y.shape).to(dev)
        (y * probe).sum().backward()
        worst, num, den, mass_err, med = 0.0, 0.0, 0.0, 0.0, 0.0
        for key, par in model.named_parameters():
            if not par.requires_grad:
                continue
            got = par.grad.detach().cpu().double().reshape(-1)
            full_scale = float(got.abs().max())
            full = "%s/grad/%s" % (name, key)
            if full in g.files:
                ref = torch.from_numpy(g[full]).double().reshape(-1)
            else:
                ref = torch.from_numpy(g["%s/gradhead/%s" % (name, key)]).double()
                mass = float(g["%s/gradabs/%s" % (name, key)])
                mass_err = max(mass_err, abs(float(got.abs().sum()) - mass) / mass)
                got = got[:512]
            err = (got - ref).abs()
            worst = max(worst, float(err.max()) / max(full_scale, 1e-12))
            med = max(med, float(err.median()) / max(full_scale, 1e-12))
            num += float((err ** 2).sum()); den += float((ref ** 2).sum())
        print("%-10s %-7s worst max/scale %.2e  worst median/scale %.2e  rel L2 %.2e  |g| sum err %.2e"
              % (name, mode, worst, med, (num / den) ** 0.5, mass_err), flush=True)
