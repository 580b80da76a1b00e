"""Cycle counts of the phases of every step of the fused forward(train) / backward-data kernels:
init (bias -> accumulators), K loops (+ feature bursts), next-step weight prefetch, epilogue loop,
mask store.  Build the instrumented library first:
    python scripts/probes/make_dbg_library.py epilogue
    FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_dbg.so python scripts/probes/epilogue_timeline.py"""
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
    out.backward(torch.randn_like(out))
    torch.cuda.synchronize()
    nb = lib.ffn_dbg_read(buf, 1); bwd = list(buf[:nb])
names = ["init", "Kloops", "prefetch", "oloop", "maskst"]
for name, t, steps in (("fwd", fwd, 3), ("bwd", bwd, 3)):
    per = 6 * steps
    for blk in range(1, min(3, len(t) // per)):
        row = t[blk * per:(blk + 1) * per]
        for st in range(steps):
            r = row[6 * st:6 * st + 6]
            print(name, "block", blk, "step", st, {k: r[i + 1] - r[i] for i, k in enumerate(names)})
