"""bf16x6 chain kernels of the full NeRF (two waves per SIMD, csrc/mlp_bf16_ws.hip) at 65 536 x 64 samples:
inference, training forward and backward data, ms.  For A/Bs of library builds (FFN_HIP_LIBRARY)."""
import ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import fourier_feature_nets_amd as ffn
from fourier_feature_nets_amd import mlp_engine as me
dev = torch.device("cuda:0")
torch.manual_seed(20080524)
model = ffn.NeRF(8, 256, 9, 10, 3, 4, [4], True).to(dev)
prog = model.program()
n = 65536 * 64
x = torch.rand(n, 3, device=dev) * 2 - 1
v = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=1)
buf = torch.zeros((prog.saved_floats(n),), dtype=torch.float32, device=dev)
d_logits = torch.randn(n, 4, device=dev) / n
prog.forward(x, v, buf, precision="bf16x6")
ws = prog.workspace(n)
_, masks = prog._split_saved(buf, n)


def bwd():
    me._call("ffn_mlp_backward_data_bf16x6", ctypes.byref(prog.bwd_x6), me._dev(prog.packed_x6_bwd, torch.int16),
             me._dev(d_logits), me.c_i64(n), me._dev(masks), me._dev(ws.dz))


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 3)


out = {"library": os.environ.get("FFN_HIP_LIBRARY", "in-tree"), "organisation": prog.x6_organisation()}
for rnd in range(2):
    out.setdefault("inference_ms", []).append(timeit(lambda: prog.forward(x, v, None, precision="bf16x6")))
    out.setdefault("train_forward_ms", []).append(timeit(lambda: prog.forward(x, v, buf, precision="bf16x6")))
    out.setdefault("backward_data_ms", []).append(timeit(bwd))
print(json.dumps(out))
