"""Per-output-tile cycle stamps of the epilogue loop of the fused forward(train) kernel
(make_dbg_library.py tiles).  Prints the deltas between consecutive stamps of one block."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.getcwd())
import fourier_feature_nets_amd as ffn
from fourier_feature_nets_amd import _lib
dev = torch.device("cuda:0")
model = ffn.PositionalFourierMLP(3, 4, 5.5).to(dev)
n = 65536 * 64
x = torch.rand(n, 3, device=dev) * 2 - 1
lib = _lib.load()
buf = (ctypes.c_longlong * 8192)()
for it in range(2):
    lib.ffn_dbg_read(buf, 1)
    out = model(x)
    torch.cuda.synchronize()
    nf = lib.ffn_dbg_read(buf, 1); fwd = list(buf[:nf])
# stamps per step: 1 (entry) + 1 (K loops done) + 8 tiles + 1 (end) = 11; 3 steps per block
per = 11 * 3
for blk in range(1, min(3, len(fwd) // per)):
    row = fwd[blk * per:(blk + 1) * per]
    for st in range(3):
        r = row[11 * st:11 * st + 11]
        print("fwd block", blk, "step", st, "kloops", r[1] - r[0], "tiles", [r[i + 1] - r[i] for i in range(1, 10)], "total", r[10] - r[0])
