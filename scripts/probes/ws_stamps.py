"""Prints the phase timeline of the WS forward kernel (variant library with cycle stamps)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fourier_feature_nets_amd as ffn
from fourier_feature_nets_amd import _lib
dev = torch.device("cuda:0")
model = (ffn.PositionalFourierMLP(3, 4, 5.5) if len(sys.argv) < 2 or sys.argv[1] == "tiny" else ffn.NeRF(8, 256, 9, 10, 3, 4, [4], True)).to(dev)
prog = model.program()
n = 1 << 22
x = torch.rand(n, 3, device=dev) * 2 - 1
v = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=1) if model.use_view else None
for _ in range(2):
    prog.forward16(x, v)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 512)()
rc = lib.ffn_debug_read_stamps(buf)
st = np.array(buf[:], dtype=np.int64).reshape(8, 64)
names = []
for w in (0, 3, 4, 7):
    row = st[w]
    k = int(np.argmax(row == 0)) if (row == 0).any() else 64
    d = np.diff(row[:k])
    print("wave", w, "total", int(row[k - 1] - row[0]), "deltas", d.tolist())
