"""One-off: cycle timestamps of one wave at every step boundary of the fused training forward
and backward-data kernels (debug build of the library with s_memtime hooks, see DESIGN.md;
run with FFN_HIP_LIBRARY=scripts/probes/variants/libffn_dbg.so)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fourier_feature_nets_amd as ffn
from fourier_feature_nets_amd import _lib

dev = torch.device("cuda:0")
model = ffn.PositionalFourierMLP(3, 4, 5.5).to(dev) if len(sys.argv) < 2 or sys.argv[1] == "tiny" \
    else ffn.NeRF(8, 256, 9, 10, 3, 4, [4], True).to(dev)
n = 65536 * 64
x = torch.rand(n, 3, device=dev) * 2 - 1
v = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=1)
lib = _lib.load()
buf = (ctypes.c_longlong * 8192)()
for it in range(2):
    lib.ffn_dbg_read(buf, 1)
    out = model(x, v) if model.use_view else model(x)
    torch.cuda.synchronize()
    nf = lib.ffn_dbg_read(buf, 1)
    fwd = list(buf[:nf])
    out.backward(torch.randn_like(out))
    torch.cuda.synchronize()
    nb = lib.ffn_dbg_read(buf, 1)
    bwd = list(buf[:nb])


def report(name, t, steps):
    per = 3 * steps
    print(name, "steps/block", steps, "stamps", len(t))
    for blk in range(min(3, len(t) // per)):
        row = t[blk * per:(blk + 1) * per]
        loops = [row[3 * i + 1] - row[3 * i] for i in range(steps)]
        epis = [row[3 * i + 2] - row[3 * i + 1] for i in range(steps)]
        print("  block %d: init+K-loop %s  epilogue %s  total %d" % (blk, loops, epis, row[-1] - row[0]))
    if len(t) >= 2 * per:
        print("  block period", t[per] - t[0])


prog = model.program()
report("forward(train)", fwd, prog.fwd.num_steps)
report("backward-data", bwd, prog.bwd.num_steps)
