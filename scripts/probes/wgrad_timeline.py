"""One-off: per-block cycle counts of the wgrad unit kernel for an encoding unit and a slab unit
(debug library: thread 0 of every workgroup logs the cycle counter at each block start into the
buffer passed as `views`)."""
import os, sys, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fourier_feature_nets_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libffn_dbg.so")
import fourier_feature_nets_amd as ffn
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ffn.PositionalFourierMLP(3, 4, 5.5).to(dev)
n = 65536 * 64
x = torch.rand(n, 3, device=dev) * 2 - 1
prog = model.program()
saved = torch.empty(prog.saved_floats(n), device=dev)
logits = prog.forward(x, None, saved)
dl = torch.randn(n, 4, device=dev) * 1e-6
grads = torch.empty(prog.num_grad_floats, device=dev)
dbg = torch.zeros(256 * 64 * 2 + 16, dtype=torch.float32, device=dev)   # viewed as u64 [256][64]
for _ in range(2):
    dbg.zero_()
    prog.backward(dl, x, dbg, saved, grads)
torch.cuda.synchronize()
t = dbg[:256 * 64 * 2].view(torch.int64).view(256, 64).cpu()
for wg in (0, 70, 140, 250):
    row = t[wg]
    d = (row[1:40] - row[:39]).tolist()
    print("wg %3d: cycles/block min %d median %d max %d   first 8: %s" % (wg, min(d), sorted(d)[len(d)//2], max(d), d[:8]))
