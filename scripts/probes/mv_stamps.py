"""Cycle stamps of the matrix-waves / vector-waves bf16x6 forward kernel (csrc/mlp_bf16_mv.hip, -DMV_STAMPS):
one pass of workgroup 0 as seen by its matrix wave 0 and its vector wave 4.

    python scripts/probes/mv_stamps.py build      # here (no GPU): scripts/probes/variants/libffn_mvstamps.so
    FFN_HIP_LIBRARY=$PWD/scripts/probes/variants/libffn_mvstamps.so python scripts/probes/mv_stamps.py run [--train]
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "scripts", "probes", "variants", "libffn_mvstamps.so")

NAMES = {1: "pass start", 2: "after P0", 3: "generated segment 0", 4: "after P1", 9: "features: K loops start",
         10: "features: pair trips done", 11: "K(A) of the last segment done", 12: "A handed over, after S2",
         13: "K(B) done", 14: "B handed over, after S4", 20: "hidden: start", 21: "hidden: K(A) done",
         22: "hidden: after S2", 23: "hidden: K(B) done", 24: "hidden: after S4", 30: "  barrier in the stream: arrive",
         31: "  barrier in the stream: leave", 32: "  feature segment (8 K blocks x 2 tiles, 192 matrix instructions)", 40: "before R1", 41: "after R1", 50: "generated a segment",
         51: "after F", 60: "before S2", 61: "after S2", 62: "epilogue A computed", 63: "after S3",
         64: "X stores of A", 65: "after S3b", 66: "after S4", 67: "epilogue B + X stores", 70: "before S1", 71: "after S1",
         80: "backward step 0: start", 81: "tile A's two units done", 82: "A handed over, after S2",
         83: "tile B's two units done", 84: "after S3", 85: "B handed over, after S3b"}


def build(extra=(), out=OUT):
    sys.path.insert(0, ROOT)
    from fourier_feature_nets_amd import build as b
    b.build_library()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    obj = os.path.join(b.CSRC, "build", "mlp_bf16_mv_stamps.o")
    subprocess.run([b._hipcc()] + b.COMMON + b.SOURCES["mlp_bf16_mv.hip"] + ["-DMV_STAMPS"] + list(extra) + ["-c",
                   os.path.join(b.CSRC, "mlp_bf16_mv.hip"), "-o", obj], check=True, capture_output=True)
    objects = [os.path.join(b.CSRC, "build", n.replace(".hip", ".o")) for n in b.SOURCES if n != "mlp_bf16_mv.hip"]
    subprocess.run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj] + objects, check=True)
    print(out)


def run(train):
    sys.path.insert(0, ROOT)
    import torch
    import fourier_feature_nets_amd as ffn
    from fourier_feature_nets_amd import _lib
    dev = torch.device("cuda:0")
    torch.manual_seed(7)
    layers = int(os.environ.get("MV_LAYERS", "3"))
    model = ffn.PositionalFourierMLP(3, 4, 5.5, num_layers=layers, num_channels=256, embedding_size=256).to(dev)
    prog = model.program()
    n = 4194304
    x = torch.rand(n, 3, device=dev) * 2 - 1
    buf = torch.zeros((prog.saved_floats(n),), dtype=torch.float32, device=dev) if train else None
    for _ in range(2):
        prog.forward(x, None, buf, precision="bf16x6")
    torch.cuda.synchronize()
    if "--bwd" in sys.argv:                        # the backward-data kernel's stamps overwrite the forward's
        from fourier_feature_nets_amd import mlp_engine as me
        d_logits = torch.randn(n, 4, device=dev) / n
        ws = prog.workspace(n)
        _, masks = prog._split_saved(buf, n)
        for _ in range(2):
            me._call("ffn_mlp_backward_data_bf16x6", ctypes.byref(prog.bwd_x6), me._dev(prog.packed_x6_bwd, torch.int16),
                     me._dev(d_logits), me.c_i64(n), me._dev(masks), me._dev(ws.dz))
        torch.cuda.synchronize()
    lib = _lib.load() if hasattr(_lib, "load") else ctypes.CDLL(os.environ["FFN_HIP_LIBRARY"])
    host = (ctypes.c_longlong * 2048)()
    rc = lib.ffn_debug_mv_stamps(host)
    assert rc == 0, rc
    for role, name in ((0, "matrix wave 0"),) if "--bwd" in sys.argv else ((0, "matrix wave 0"), (1, "vector wave 4")):
        print("== %s (%s)" % (name, "training forward" if train else "inference"))
        rows = []
        for i in range(510):
            ident, t = host[role * 1024 + 2 * i], host[role * 1024 + 2 * i + 1]
            if ident == 0:
                break
            rows.append((ident, t))
        t0 = rows[0][1]
        prev = t0
        for ident, t in rows:
            print("%8d  +%6d  %s" % (t - t0, t - prev, NAMES.get(ident, str(ident))))
            prev = t


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
        for ko in sys.argv[2:]:                    # timing-only knock-outs: build MV_KO_PAIR_W MV_KO_PAIR_X
            build(["-D" + ko], OUT.replace("mvstamps", "mvstamps_" + ko.lower()))
    else:
        run("--train" in sys.argv)
