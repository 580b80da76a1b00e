"""One-off: cycle timestamps of one wave at every step boundary of the fused forward
(debug build of the library with __builtin_readcyclecounter hooks)."""
import ctypes, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fourier_feature_nets_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libffn_dbg.so")
from fourier_feature_nets_amd.mlp_engine import DenseSpec, EncodingSpec, MlpProgram
from fourier_feature_nets_amd.ops import _dev, _stream
from fourier_feature_nets_amd._lib import c_i64
from oracle import ffn_oracle as orc
dev = torch.device("cuda:0")
C = 256
for model in ("raw", "positional"):
    if model == "positional":
        b = orc.positional_b_values(5.5, 256, 3).to(dev); a = torch.ones(b.shape[1], device=dev); first = 2 * b.shape[1]
    else:
        b, a, first = None, None, 3
    dims = [(C, first)] + [(C, C)] * 4 + [(4, C)]
    layers = []
    for i, (o, k) in enumerate(dims):
        lin = torch.nn.Linear(k, o); last = i == len(dims) - 1
        layers.append(DenseSpec(lin.weight.detach().to(dev), lin.bias.detach().to(dev), 0 if i == 0 else k, 0 if i == 0 else None, not last, (0, 4) if last else None))
    prog = MlpProgram([EncodingSpec(b, a, math.pi, False, dev)], layers, dev)
    prog.pack()
    n = 65536 * 64
    x = torch.rand(n, 3, device=dev) * 2 - 1
    logits = torch.empty((n, 4), device=dev)
    dbg = torch.zeros(4096, dtype=torch.int64, device=dev)
    for _ in range(2):
        _lib.call("ffn_mlp_forward", ctypes.byref(prog.fwd), _dev(prog.packed_fwd), _dev(prog.bias_buf), _dev(x), _dev(None), c_i64(n), _dev(logits), _dev(None), ctypes.c_void_p(dbg.data_ptr()), _stream())
    torch.cuda.synchronize()
    t = dbg.cpu().tolist()
    t = [v for v in t if v != 0][:2 * len(dims) * 3]
    print(model, "steps", len(dims))
    per = 2 * len(dims)
    for blk in range(len(t) // per):
        row = t[blk * per:(blk + 1) * per]
        loops = [row[2 * i + 1] - row[2 * i] for i in range(len(dims))]
        gaps = [row[2 * i + 2] - row[2 * i + 1] for i in range(len(dims) - 1)]
        print("  block %d: K-loop cycles %s   epilogue+handoff cycles %s   total %d" % (blk, loops, gaps, row[-1] - row[0]))
