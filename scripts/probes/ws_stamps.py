"""Prints the phase timeline of a two-waves-per-SIMD forward kernel (variant library with cycle
stamps, scripts/probes/ws_variants/stamps_r5.py):
    FFN_HIP_LIBRARY=scripts/probes/variants/libffn_ws_stamps.so python scripts/probes/ws_stamps.py [tiny|nerf|mlp8] [bf16x3|bf16x6] [train]
Stamps (s_memtime, 100 MHz constant clock? -- no: the shader clock counter) of workgroup 0, its last pass:
per step: start, after the activation K loop; per feature segment: start, between run / generate,
end; then before / after the "all consumed" barrier and before / after the "filled" barrier."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fourier_feature_nets_amd as ffn
from fourier_feature_nets_amd import _lib
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
train = len(sys.argv) > 3 and sys.argv[3] == "train"
model = {"tiny": lambda: ffn.PositionalFourierMLP(3, 4, 5.5), "nerf": lambda: ffn.NeRF(8, 256, 9, 10, 3, 4, [4], True),
         "mlp8": lambda: ffn.MLP(3, 4, num_layers=8, num_channels=256)}[name]().to(dev)
prog = model.program()
n = 1 << 22
x = torch.rand(n, 3, device=dev) * 2 - 1
v = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=1) if model.use_view else None
saved = torch.empty((prog.saved_floats(n),), dtype=torch.float32, device=dev) if train else None
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(3):
    if it == 2:
        e0.record()
    if mode == "bf16x3" and not train:
        prog.forward16(x, v)
    else:
        prog.forward(x, v, saved, precision=mode)
e1.record()
torch.cuda.synchronize()
print(name, mode, "train" if train else "infer", "kernel %.3f ms" % e0.elapsed_time(e1))
lib = _lib.load()
buf = (ctypes.c_ulonglong * 1024)()
rc = lib.ffn_debug_read_stamps(buf)
st = np.array(buf[:], dtype=np.int64).reshape(16, 64)
for w in (0, 3, 4, 7):
    row = st[w]
    k = int(np.argmax(row == 0)) if (row == 0).any() else 64
    d = np.diff(row[:k])
    print("wave", w, "stamps", k, "total", int(row[k - 1] - row[0]), "deltas", d.tolist())
