"""Split-bf16 inference kernel vs the exact-f32 inference kernel at the bench shape."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fourier_feature_nets_amd as ffn

dev = torch.device("cuda:0")
out = {}
for name, make, view in (("tiny", lambda: ffn.PositionalFourierMLP(3, 4, 5.5), False),
                         ("nerf", lambda: ffn.NeRF(8, 256, 9, 10, 3, 4, [4], True), True)):
    torch.manual_seed(20080524)
    model = make().to(dev)
    prog = model.program()
    n = 65536 * 64
    x = torch.rand(n, 3, device=dev) * 2 - 1
    v = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=1) if view else None
    flops = 2 * sum(sp.out * sp.ld for sp in prog.layers) * n
    res = {}
    for mode, fn in (("f32", lambda: prog.forward(x, v, None)), ("bf16x3", lambda: prog.forward16(x, v))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            y = fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        res[mode] = {"ms": round(ms, 3), "algorithmic_TFLOPs": round(flops / ms / 1e9, 1), "y": y}
    err = float((res["f32"]["y"] - res["bf16x3"]["y"]).abs().max())
    scale = float(res["f32"]["y"].abs().max())
    for m in res:
        del res[m]["y"]
    out[name] = {"samples": n, **res, "speedup": round(res["f32"]["ms"] / res["bf16x3"]["ms"], 2),
                 "max_abs_logit_error": err, "max_abs_logit": scale}
print(json.dumps(out, indent=1))
