"""What a pure read stream gets from HBM here (reference point for the slab-bound kernels)."""
import torch
x = torch.empty(2 * 1024 ** 3, dtype=torch.float32, device="cuda").normal_()   # 8 GiB
for name, fn in (("sum (f32)", lambda: x.sum()), ("max", lambda: x.max()), ("abs().sum via norm1", lambda: torch.linalg.vector_norm(x, 1))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("%-24s %.3f ms  %.2f TB/s" % (name, ms, x.numel() * 4 / ms / 1e9))
