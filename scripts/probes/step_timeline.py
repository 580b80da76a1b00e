"""One-off: cycle timestamps of one wave at every step boundary of the fused training forward
and backward-data kernels (instrumented build: python scripts/probes/make_dbg_library.py steps;
run with FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_dbg.so)."""
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
    """Stamps per step: start, (negative) one per K segment start, end of K loops, end of epilogue."""
    print(name, "stamps", len(t))
    blocks, cur = [], []
    ends = 0
    for v in t:
        cur.append(v)
        if v > 0 and len(cur) > 1 and cur[-2] > 0 and sum(1 for c in cur if c > 0) % 3 == 0:
            ends += 1
            if ends == steps:
                blocks.append(cur)
                cur, ends = [], 0
    for blk in blocks[:3]:
        out, i = [], 0
        pos = [abs(v) for v in blk]
        marks = ["S" if v > 0 else "k" for v in blk]
        print("  block: " + " ".join("%s+%d" % (m, b - a) for m, a, b in zip(marks[1:], pos, pos[1:])),
              " total", pos[-1] - pos[0])


prog = model.program()
report("forward(train)", fwd, prog.fwd.num_steps)
report("backward-data", bwd, prog.bwd.num_steps)
