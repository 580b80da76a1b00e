"""Times the weight-gradient kernels ALONE, back to back (slabs filled once), against the same
kernels inside the forward -> backward-data -> weight-gradient sequence."""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import torch
import fourier_feature_nets_amd as ffn
from fourier_feature_nets_amd import _lib
dev = torch.device("cuda:0")
torch.manual_seed(1)
n = 65536 * 64
model = ffn.PositionalFourierMLP(3, 4, 5.5).to(dev)
prog = model.program()
x = torch.rand(n, 3, device=dev) * 2 - 1
saved = torch.empty((prog.saved_floats(n),), dtype=torch.float32, device=dev)
grads = torch.empty((prog.num_grad_floats,), dtype=torch.float32, device=dev)
d_logits = torch.randn(n, 4, device=dev) / n
orig = _lib.call
for mode in ("f32", "bf16x3"):
    prog.forward(x, None, saved, precision=mode)
    captured = {}
    def hook(name, *a):
        if "wgrad_units" in name:
            captured["call"] = (name, a)
        return orig(name, *a)
    _lib.call = hook
    prog.backward(d_logits, x, None, saved, grads, precision=mode)
    _lib.call = orig
    name, a = captured["call"]
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
    e[0].record()
    for i in range(6):
        orig(name, *a)
        e[i + 1].record()
    torch.cuda.synchronize()
    print(mode, name, "alone, back to back (ms):", [round(e[i].elapsed_time(e[i + 1]), 3) for i in range(6)])
